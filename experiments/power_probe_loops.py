#!/usr/bin/env python3
"""Workloads for experiments/power_trace.py that hold ONE regime for several seconds:
    python experiments/power_probe_loops.py mfma [seconds]     register-only fp16 MFMA loops (d3d_probe_machine(0))
    python experiments/power_probe_loops.py stage [seconds]    the k-loop's L2 -> LDS staging stream alone (d3d_probe_machine(1))
    python experiments/power_probe_loops.py b1 [seconds]       9-step samplings at B = 1 (every launch under-fills the chip)
prints one JSON line with the rate it held."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "mfma"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda", 0)
if what in ("mfma", "stage"):
    import ctypes as C
    from diff3dhpe_amd import _lib
    vals, t0 = [], time.time()
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        while time.time() - t0 < secs:
            r = C.c_float(0.0)
            _lib.check(_lib.lib().d3d_probe_machine(0 if what == "mfma" else 1, 400.0, C.byref(r), st))
            vals.append(r.value)
    print(json.dumps({"value": round(sum(vals) / len(vals), 1), "unit": "TFLOP/s fp16 MFMA" if what == "mfma" else "GB/s L2->LDS", "n": len(vals)}))
else:
    import diff3dhpe_amd as d3d
    from diff3dhpe_amd.spec import DenoiserConfig
    from diff3dhpe_amd.synth import synth_state_dict, synth_inputs_rows
    cfg = DenoiserConfig(num_frame=243, embed_dim=512, depth=8)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=243, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8, mlp_ratio=2., qkv_bias=True,
                                      qk_scale=None, drop_path_rate=0.1, with_time_emb=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
    net.precision = "f16x3"
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=9, loss_type="l2", clip_denoised=True, beta_schedule="cosine",
                                 ddim_sampling_eta=0.0, clipLoss=True).eval().to(dev)
    eng = diff._engine(dev)
    inp = synth_inputs_rows(0, 1, 243, seed=42)
    x2d, nz = torch.from_numpy(inp["x2d"]).to(dev), torch.from_numpy(inp["noise"]).to(dev)
    eng.ddim_sample(x2d, nz); torch.cuda.synchronize()
    n, t0 = 0, time.time()
    while time.time() - t0 < secs:
        for _ in range(20):
            eng.ddim_sample(x2d, nz)
        torch.cuda.synchronize(); n += 20
    print(json.dumps({"value": round((time.time() - t0) / n * 1e3, 3), "unit": "ms per 9-step sampling at B = 1", "n": n}))
