"""How exact is the fp32 accumulation of the bf16 MFMA path?  Same bf16-representable operands through the bf16 GEMM, the F16X3
GEMM and the fp32-MFMA GEMM (all exact in their inputs), each against fp64 math."""
import sys, torch
sys.path.insert(0, ".")
from diff3dhpe_amd.engine import op_linear
torch.manual_seed(0)
for M, N, K in ((512, 512, 512), (512, 512, 1024)):
    A = torch.randn(M, K).to(torch.bfloat16).float()
    W = (torch.randn(N, K) / K ** 0.5).to(torch.bfloat16).float()
    R = torch.zeros(M, N)
    ref = A.double() @ W.double().T
    for prec in ("bf16", "f16x3", "fp32"):
        out = op_linear(A.cuda(), W.cuda(), torch.zeros(N).cuda(), R.cuda(), epi="residual", precision=prec).cpu().double()
        d = (out - ref).abs()
        print(f"M={M} N={N} K={K} [{prec}]: max-abs {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e} (|ref| rms {ref.pow(2).mean().sqrt():.3f})")
    cpu = (A @ W.T).double()
    print(f"   CPU fp32 sgemm: max-abs {(cpu - ref).abs().max():.3e} rms {(cpu - ref).pow(2).mean().sqrt():.3e}")
