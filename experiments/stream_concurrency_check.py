#!/usr/bin/env python3
"""Bitwise comparison of one B=64 sampling against its two halves run (a) one after the other, (b) concurrently on two
streams, several times: any difference between (a) and the full batch is a batch dependence, any difference that only
(b) shows is a race that needs co-running kernels to appear."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs
B, T, S = int(os.environ.get("B", 64)), 243, int(os.environ.get("S", 3))
dev = torch.device("cuda:0")
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()}
def make():
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=8)
    net.load_state_dict(sd); net.precision = "f16x3"
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0).eval().to(dev)
    return diff, diff._engine(dev)
keep = [make(), make()]
engs = [k[1] for k in keep]
inp = synth_inputs(B, T, seed=42)
x2d = torch.from_numpy(inp["x2d"]).to(dev); noise = torch.from_numpy(inp["noise"]).to(dev)
h = B // 2
full = engs[0].ddim_sample(x2d, noise).clone()
full2 = engs[0].ddim_sample(x2d, noise).clone()
print("full batch repeatable:", bool((full == full2).all()))
seq = torch.cat([engs[0].ddim_sample(x2d[:h].contiguous(), noise[:h].contiguous()), engs[1].ddim_sample(x2d[h:].contiguous(), noise[h:].contiguous())])
d = seq != full
print("sequential halves vs full: differing elements", int(d.sum()), "max", (seq - full).abs().max().item(), "samples", d.flatten(1).any(1).nonzero().flatten().tolist()[:10])
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for it in range(4):
    outs = []
    torch.cuda.synchronize()
    for i, (lo, hi) in enumerate([(0, h), (h, B)]):
        with torch.cuda.stream(streams[i]):
            outs.append(engs[i].ddim_sample(x2d[lo:hi].contiguous(), noise[lo:hi].contiguous()))
    torch.cuda.synchronize()
    con = torch.cat(outs)
    d = con != seq
    print(f"concurrent run {it} vs sequential halves: differing elements {int(d.sum())} max {(con - seq).abs().max().item():.3e} samples {d.flatten(1).any(1).nonzero().flatten().tolist()[:10]}")
