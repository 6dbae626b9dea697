#!/usr/bin/env python3
"""Round 5: what the default range-guard read costs a caller of the PYTHON API at small batches, where a call is short enough for the
host's launch work to matter.  GaussianDiffusion.forward() waits on its call's ticket before it returns (the result is about to be used), so
back-to-back calls no longer overlap their launch work with the previous call's kernels.  B sequences, T = 243, 9 steps, back-to-back calls:
    guard on (default) / guard off (net.range_check = False), eager / hipGraph replay.
    python experiments/guard_latency.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs_rows

T, S = 243, 9
dev = torch.device("cuda", 0)
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8, mlp_ratio=2., qkv_bias=True,
                                  qk_scale=None, drop_path_rate=0.1)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True, beta_schedule="cosine",
                             ddim_sampling_eta=0.0, clipLoss=True).eval().to(dev)
eng = diff._engine(dev)
if os.environ.get("STREAMS"):
    eng.set_option("streams", int(os.environ["STREAMS"]))
    print("streams", os.environ["STREAMS"])
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64]:
    inp = synth_inputs_rows(0, B, T, seed=42)
    x2d, nz = torch.from_numpy(inp["x2d"]).to(dev), torch.from_numpy(inp["noise"]).to(dev)
    z = torch.zeros_like(nz)
    row = []
    for graph in (False, True):
        eng.set_graph_mode(graph)
        for guard in (True, False):
            net.range_check = guard
            n = max(3, min(30, int(1500 / (8 * B + 20))))
            for _ in range(2):
                diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / n * 1e3)
    eng.set_graph_mode(False)
    net.range_check = True
    print(f"B={B:3d}  ms per forward() call, back to back:  eager guard on {row[0]:8.2f} / off {row[1]:8.2f} ({row[0] / row[1] - 1:+.1%})   "
          f"graph guard on {row[2]:8.2f} / off {row[3]:8.2f} ({row[2] / row[3] - 1:+.1%})", flush=True)
