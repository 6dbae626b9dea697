#!/usr/bin/env python3
"""The sampler's result must not depend on what the (caller-owned, uninitialised) workspace holds on entry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs
T, B, S = 27, 3, 3
dev = torch.device("cuda:0")
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=8)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
for prec in ("f16x3", "fp32"):
    net.precision = prec
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0).eval().to(dev)
    eng = diff._engine(dev)
    inp = synth_inputs(B, T, seed=42)
    x2d = torch.from_numpy(inp["x2d"]).to(dev); noise = torch.from_numpy(inp["noise"]).to(dev)
    ws = eng._workspace(B)
    ws.zero_()
    ref = eng.ddim_sample(x2d, noise).clone()
    for name, fill in (("0xFF (NaN)", lambda: ws.fill_(255)), ("random bytes", lambda: ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device=dev))),
                       ("0x7C (fp16 inf / big)", lambda: ws.fill_(0x7C))):
        fill()
        y = eng.ddim_sample(x2d, noise)
        d = y != ref
        print(f"{prec} workspace = {name}: {int(d.sum())} differing elements, max diff {(y - ref).abs().max().item():.3e}, nan {int(torch.isnan(y).sum())}", flush=True)
