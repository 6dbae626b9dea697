#!/usr/bin/env python3
"""Round 5: would MORE than two concurrent part-batches help?  (The proj / fc2 epilogues run in chip-wide lock step and their HBM traffic
comes in bursts -- NOTES section 0.14; two streams de-phase two halves.)  Two engines with the same weights, each sampling HALF of the
headline batch on its own torch stream with its own internal two-stream split = four quarter-batches in flight, against the shipped
form (one engine, the whole batch, two internal streams).  hipGraph replay on both sides so that launch order does not matter.
    python experiments/four_way.py [B] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs_rows

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T, S = 243, 9
dev = torch.device("cuda", 0)
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()}


def make():
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8, mlp_ratio=2.,
                                      qkv_bias=True, qk_scale=None, drop_path_rate=0.1)
    net.load_state_dict(sd)
    net.precision = "f16x3"
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0, clipLoss=True).eval().to(dev)
    return diff._engine(dev), diff, net


inp = synth_inputs_rows(0, B, T, seed=42)
x2d, nz = torch.from_numpy(inp["x2d"]).to(dev), torch.from_numpy(inp["noise"]).to(dev)
e0, d0, n0 = make()
e1, d1, n1 = make()
for e in (e0, e1):
    e.set_option("streams", 2)
    e.set_graph_mode(True)
h = B // 2
xa, xb, na, nb = x2d[:h].contiguous(), x2d[h:].contiguous(), nz[:h].contiguous(), nz[h:].contiguous()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def whole():
    return e0.ddim_sample(x2d, nz)


def four():
    with torch.cuda.stream(s1):
        a = e0.ddim_sample(xa, na)
    with torch.cuda.stream(s2):
        b = e1.ddim_sample(xb, nb)
    return a, b


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

for r in range(reps):
    tw, yw = timed(whole, 2)
    tf, (ya, yb) = timed(four, 2)
    same = torch.equal(torch.cat([ya, yb]), yw)
    print(f"B={B}: one engine, two streams {tw:8.2f} ms ({B / tw * 1e3:6.2f} seq/s) | two engines x two streams {tf:8.2f} ms ({B / tf * 1e3:6.2f} seq/s)  bit-identical {same}", flush=True)
