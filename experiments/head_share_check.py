#!/usr/bin/env python3
"""The regression-head kernel alone (d3d_op_head) on fixed rows, repeated, as N processes at once on one GPU: every result must
equal the first bit for bit.  Prints which rows / values moved.   python experiments/head_share_check.py [procs] [reps] [rows]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import diff3dhpe_amd as d3d
    from diff3dhpe_amd.spec import DenoiserConfig
    from diff3dhpe_amd.synth import synth_state_dict
    reps, rows, tag = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dev = torch.device("cuda:0")
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=1)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=27, embed_dim=512, depth=1)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
    net.precision = "f16x3"
    eng = net.engine_for(dev)
    torch.manual_seed(5)
    X = torch.randn(rows, 512, device=dev)
    ref = eng.head(X).clone()
    bad = 0
    for it in range(reps):
        y = eng.head(X)
        if not torch.equal(y, ref):
            d = (y != ref)
            r = d.any(1).nonzero().flatten().tolist()
            bad += 1
            if bad <= 8:
                i = r[0]
                print(f"{tag} rep {it}: {int(d.sum())} values in {len(r)} rows differ; rows {r[:12]}{'...' if len(r) > 12 else ''}; "
                      f"row {i}: got {y[i].tolist()} expected {ref[i].tolist()}", flush=True)
    print(f"{tag}: {bad} of {reps} launches differ", flush=True)
    sys.exit(0)
procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1377
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(reps), str(rows), f"p{i}"], cwd=ROOT) for i in range(procs)]
sys.exit(max(p.wait() for p in ps))
