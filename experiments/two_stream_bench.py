#!/usr/bin/env python3
"""Does running two half-batches on two HIP streams overlap the HBM-bound kernels of one with the MFMA-bound GEMMs of the
other?  Samples are independent through the whole DDIM loop, so the split is a pure re-scheduling (bit-identical output).

    python experiments/two_stream_bench.py [B] [T] [S] [reps]          PARTS="32,32;22,21,21;16,16,16,16" for other splits
The engine does the 2-way equal split by itself since round 3 ("streams" = 2, the default); here every sub-call runs with
"streams" = 1 and the script does the splitting, so that other ratios and 3- / 4-way splits can be compared.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 243
S = int(sys.argv[3]) if len(sys.argv) > 3 else 9
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()}


def make():
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=8)
    net.load_state_dict(sd)
    net.precision = "f16x3"
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0).eval().to(dev)
    return diff, diff._engine(dev)


inp = synth_inputs(B, T, seed=42)
x2d = torch.from_numpy(inp["x2d"]).to(dev)
noise = torch.from_numpy(inp["noise"]).to(dev)
keep = [make() for _ in range(4)]
engs = [k[1] for k in keep]
for e in engs:
    e.set_option("streams", 1)
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def one_stream():
    return engs[0].ddim_sample(x2d, noise)


def two_streams(split):
    return multi([split, B - split])


def multi(parts):
    outs = []
    cur = torch.cuda.current_stream(dev)
    bounds, lo = [], 0
    for p_ in parts:
        bounds.append((lo, lo + p_))
        lo += p_
    for i, (lo, hi) in enumerate(bounds):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs.append(engs[i].ddim_sample(x2d[lo:hi], noise[lo:hi]))
    for s in streams[:len(bounds)]:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


def timeit(fn, *a):
    fn(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t1, ref = timeit(one_stream)
print(f"one stream   B={B}: {t1 * 1e3:8.1f} ms  {B / t1:7.2f} seq/s", flush=True)
# round quantisation of the persistent GEMM walks: B=63 at T=243 is 7.95 / 15.9 / 23.8 rounds of 256 tiles, B=64 8.07 / 16.1 / 24.2
for b in (B - 1, B - 2, B - 9):
    tb, _ = timeit(lambda: engs[0].ddim_sample(x2d[:b].contiguous(), noise[:b].contiguous()))
    print(f"one stream   B={b}: {tb * 1e3:8.1f} ms  {b / tb:7.2f} seq/s", flush=True)
splits = [int(v) for v in os.environ.get("SPLITS", f"{B // 2},{B - 1},{B - 2},{B - 9}").split(",")]
for split in splits:
    t2, out = timeit(two_streams, split)
    print(f"two streams {split}+{B - split}: {t2 * 1e3:8.1f} ms  {B / t2:7.2f} seq/s  bit-identical={bool((out == ref).all())}", flush=True)
for spec in [v for v in os.environ.get("PARTS", "").split(";") if v]:
    parts = [int(v) for v in spec.split(",")]
    assert sum(parts) == B and len(parts) <= 4
    t2, out = timeit(multi, parts)
    print(f"streams {'+'.join(map(str, parts))}: {t2 * 1e3:8.1f} ms  {B / t2:7.2f} seq/s  bit-identical={bool((out == ref).all())}", flush=True)
