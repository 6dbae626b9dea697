#!/usr/bin/env python3
"""Reference point only (never used by the product): what does the vendor fp16 GEMM (hipBLASLt via torch.matmul) reach on
the token-GEMM shapes?  Plain fp16 x fp16 -> fp16, i.e. ONE MFMA per product; F16X3 issues three."""
import torch, time
M = 264384
for name, N, K in (("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 1024, 512), ("fc2", 512, 1024)):
    A = torch.randn(M, K, device="cuda", dtype=torch.float16)
    W = (torch.rand(N, K, device="cuda") - 0.5).to(torch.float16)
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:5s} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s (fp16 in / fp16 out, 1 MFMA per product)")
