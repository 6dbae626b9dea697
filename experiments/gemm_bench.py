#!/usr/bin/env python3
"""GEMM micro-benchmark through the C ABI (d3d_op_linear_bench): token GEMM shapes of the T=243, B=64 workload.
Prints algorithmic TFLOP/s per (shape, precision, tile variant) and the max error vs fp64 on a row sample.
    python experiments/gemm_bench.py [M] [variants, e.g. 13,0,4] [diag]     diag: in-kernel stamp reports ("gemm_diag" option)
    variants: 0 = the engine's choice, 13 = 256x256 one workgroup per tile, 4 = 256x128; "bf16" as a variant runs the bf16 mode"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff3dhpe_amd.engine import op_linear_bench

M = int(sys.argv[1]) if len(sys.argv) > 1 else 264384
torch.manual_seed(0)
shapes = [("qkv", 1536, 512, "none"), ("proj", 512, 512, "residual"), ("fc1", 1024, 512, "gelu"), ("fc2", 512, 1024, "residual")]
variants = [(("bf16", 0) if v == "bf16" else ("f16x3", int(v))) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["13", "0", "4"])]
if len(sys.argv) > 3 and sys.argv[3] == "diag":
    from diff3dhpe_amd import _lib
    _lib.check(_lib.lib().d3d_engine_set_option(None, b"gemm_diag", 1))
res = {}
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda")
    W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    b = torch.rand(N, device="cuda") - 0.5
    R = torch.randn(M, N, device="cuda") if epi == "residual" else None
    idx = torch.randint(0, M, (512,), device="cuda")
    ref = A[idx].double() @ W.double().t() + b.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "residual":
        ref = ref + R[idx].double()
    for prec, var in variants:
        try:
            out, ms = op_linear_bench(A, W, b, R, epi=epi, precision=prec, variant=var, reps=10)
            err = (out[idx].double() - ref).abs().max().item()
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"{name:5s} N={N:5d} K={K:5d} {prec:6s} v{var}: {ms:8.3f} ms  {tf:7.1f} TF/s  max-err {err:.2e}", flush=True)
            res[f"{name}:{prec}:v{var}"] = {"ms": ms, "tflops": tf, "err": err}
        except Exception as e:
            print(f"{name} {prec} v{var}: FAILED {e}", flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "gemm_bench.json"), "w"), indent=1)
