#!/usr/bin/env python3
"""Eager launches vs hipGraph replay of the whole DDIM loop (BASELINE configs[3]: T=243, 50 steps), small batches."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs

T, S = 243, 50
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=8)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
net.precision = "f16x3"
diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True).eval().cuda()
eng = diff._engine(torch.device("cuda:0"))
res = {}
for B in (1, 2, 8, 32):
    inp = synth_inputs(B, T, seed=1)
    x2d, nz = torch.from_numpy(inp["x2d"]).cuda(), torch.from_numpy(inp["noise"]).cuda()
    for mode in ("eager", "graph"):
        eng.set_graph_mode(mode == "graph")
        eng.ddim_sample(x2d, nz); torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.ddim_sample(x2d, nz)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[f"B{B}:{mode}"] = {"ms": dt * 1e3, "seq_per_s": B / dt}
        print(f"T={T} S={S} B={B:3d} {mode:5s}: {dt*1e3:9.2f} ms  {B/dt:8.2f} pose-seq/s", flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "graph_bench.json"), "w"), indent=1)
