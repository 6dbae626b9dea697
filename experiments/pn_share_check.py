#!/usr/bin/env python3
"""Which elements of the post-norm GEMM's output move when a second process shares the GPU?
    python experiments/pn_share_check.py [M] [reps] & python experiments/pn_share_check.py [M] [reps]; wait
Runs d3d_op_linear_postnorm `reps` times on fixed inputs and compares every result with the first one, bitwise."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff3dhpe_amd import engine as E

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2754
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
stats = len(sys.argv) > 3 and sys.argv[3] == "planes"
N, K = 512, 1024
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g)
R = torch.randn(M, N, device="cuda", generator=g)
gam = 1 + 0.1 * torch.randn(N, device="cuda", generator=g)
bet = 0.1 * torch.randn(N, device="cuda", generator=g)
ref, st0, _ = E.op_linear_postnorm(A, W, b, R, gam, bet, with_stats=stats)
bad_runs = 0
for i in range(reps):
    y, st, _ = E.op_linear_postnorm(A, W, b, R, gam, bet, with_stats=stats)
    if stats and not torch.equal(st, st0):
        print(f"[pid {os.getpid()}] run {i}: statistics differ in {int((st != st0).any(1).sum())} rows", flush=True)
    if not torch.equal(y, ref):
        bad_runs += 1
        d = (y != ref)
        rows = d.any(1).nonzero().flatten()
        cols = d.any(0).nonzero().flatten()
        if bad_runs <= 5:
            print(f"[pid {os.getpid()}] run {i}: {int(d.sum())} elements differ; rows {rows[:12].tolist()} (n={rows.numel()}) "
                  f"cols {cols[:12].tolist()} (n={cols.numel()}) max |d| {float((y - ref).abs().max()):.3e}", flush=True)
print(f"[pid {os.getpid()}] M={M}: {bad_runs} of {reps} runs differ from the first", flush=True)
