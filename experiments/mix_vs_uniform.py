#!/usr/bin/env python3
"""The two-shape GEMM launch (variant 0 on large problems) must reproduce the uniform 256x256 launch (variant 13) bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff3dhpe_amd.engine import op_linear_bench
M = int(sys.argv[1]) if len(sys.argv) > 1 else 264384
torch.manual_seed(0)
for name, N, K, epi in [("qkv", 1536, 512, "none"), ("proj", 512, 512, "residual"), ("fc1", 1024, 512, "gelu"), ("fc2", 512, 1024, "residual")]:
    A = torch.randn(M, K, device="cuda"); W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    b = torch.rand(N, device="cuda") - 0.5
    R = torch.randn(M, N, device="cuda") if epi == "residual" else None
    o13, _ = op_linear_bench(A, W, b, R, epi=epi, precision="f16x3", variant=13, reps=1)
    o13 = o13.clone()
    for rep in range(2):
        o0, _ = op_linear_bench(A, W, b, R, epi=epi, precision="f16x3", variant=0, reps=1)
        d = o0 != o13
        rows = d.any(1).nonzero().flatten()
        print(f"{name} rep{rep}: {int(d.sum())} differing elements in {rows.numel()} rows; first {rows[:6].tolist()} last {rows[-3:].tolist()} max {(o0 - o13).abs().max().item():.3e}", flush=True)
