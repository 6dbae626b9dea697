#!/usr/bin/env python3
"""Localise a batch-dependence: rows of a GEMM / of the whole sampler must not depend on how many rows follow them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff3dhpe_amd.engine import op_linear
torch.manual_seed(0)
dev = "cuda"
for name, N, K, epi in [("qkv", 1536, 512, "none"), ("proj", 512, 512, "residual"), ("fc1", 1024, 512, "gelu"), ("fc2", 512, 1024, "residual")]:
    M2 = 2754
    A = torch.randn(M2, K, device=dev); W = (torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5
    b = torch.rand(N, device=dev) - 0.5; R = torch.randn(M2, N, device=dev)
    full = op_linear(A, W, b, R if epi == "residual" else None, epi=epi, precision="f16x3")
    ref = A.double() @ W.double().t() + b.double()
    if epi == "gelu": ref = torch.nn.functional.gelu(ref)
    if epi == "residual": ref = ref + R.double()
    print(name, "max err vs fp64", (full.double() - ref).abs().max().item())
    for M1, off in [(1377, 0), (1377, 1377), (1000, 300)]:
        part = op_linear(A[off:off + M1].contiguous(), W, b, R[off:off + M1].contiguous() if epi == "residual" else None, epi=epi, precision="f16x3")
        d = (part != full[off:off + M1])
        rows = d.any(1).nonzero().flatten()
        print(f"  rows[{off}:{off+M1}] vs full: {int(d.sum())} differing elements in {rows.numel()} rows; first rows {rows[:8].tolist()} max diff {(part - full[off:off+M1]).abs().max().item():.3e}")

# whole sampler: B=6 against its two halves (T=27, 3 DDIM steps, the configuration of tests/test_gpu_parity.py)
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs
T = int(os.environ.get("T", "27"))
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=int(os.environ.get("DEPTH", "8")))
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()}
for prec in ("f16x3", "fp32"):
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=cfg.depth)
    net.load_state_dict(sd); net.precision = prec
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=3, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0).eval().to(dev)
    eng = diff._engine(torch.device("cuda:0"))
    inp = synth_inputs(6, T, seed=42)
    x2d = torch.from_numpy(inp["x2d"]).to(dev); noise = torch.from_numpy(inp["noise"]).to(dev)
    full = eng.ddim_sample(x2d, noise).clone()
    again = eng.ddim_sample(x2d, noise).clone()
    h0 = eng.ddim_sample(x2d[:3].contiguous(), noise[:3].contiguous()).clone()
    h1 = eng.ddim_sample(x2d[3:].contiguous(), noise[3:].contiguous()).clone()
    halves = torch.cat([h0, h1])
    print(prec, "rerun identical:", bool((full == again).all()), " per-sample max |full - halves|:",
          [(full[b] - halves[b]).abs().max().item() for b in range(6)])
    t = torch.full((1,), 500.0, device=dev)
    y = noise
    f1 = eng.denoise(x2d, y, t).clone()
    g0 = eng.denoise(x2d[:3].contiguous(), y[:3].contiguous(), t).clone(); g1 = eng.denoise(x2d[3:].contiguous(), y[3:].contiguous(), t).clone()
    print(prec, "one denoise per-sample max diff:", [(f1[b] - torch.cat([g0, g1])[b]).abs().max().item() for b in range(6)])
