#!/usr/bin/env python3
"""fc2 + post-norm in one GEMM (128x512 whole-row tiles) against the plain fc2 GEMM (256x256 tiles, fp32 residual) + the row
LayerNorm kernel, at the token count of the T=243 / B=64 workload.

    python experiments/postnorm_bench.py [M] [reps] [K]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff3dhpe_amd import engine as E

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 243 * 17
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, K = 512, (int(sys.argv[3]) if len(sys.argv) > 3 else 1024)
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g)
R = torch.randn(M, N, device="cuda", generator=g)
gam = 1 + 0.1 * torch.randn(N, device="cuda", generator=g)
bet = 0.1 * torch.randn(N, device="cuda", generator=g)
tv = torch.randn(N, device="cuda", generator=g)

_, ms_plain = E.op_linear_bench(A, W, b, R, epi="residual", precision="f16x3", reps=reps)
x = torch.randn(M, N, device="cuda", generator=g)
E.op_layernorm(x, gam, bet, 1e-6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    E.op_layernorm(x, gam, bet, 1e-6)
torch.cuda.synchronize()
ms_ln = (time.perf_counter() - t0) / reps * 1e3
_, _, ms_pn32 = E.op_linear_postnorm(A, W, b, R, gam, bet, tvec=tv, reps=reps)
_, _, ms_pnpl = E.op_linear_postnorm(A, W, b, R, gam, bet, tvec=tv, with_stats=True, reps=reps)
fl = 2.0 * M * N * K
print(f"M={M}: fc2 256x256 + fp32 residual {ms_plain:.3f} ms ({fl / ms_plain * 1e-9:.0f} TFLOP/s), row LayerNorm {ms_ln:.3f} ms; "
      f"post-norm GEMM fp32-out {ms_pn32:.3f} ms, plane-out + stats {ms_pnpl:.3f} ms ({fl / ms_pnpl * 1e-9:.0f} TFLOP/s)")
