"""Are the bf16 roundings of the engine's kernels round-to-nearest-even?  Fraction of outputs that equal the RNE rounding of the
exactly computed value (fp64 math on the same operands)."""
import sys, torch
sys.path.insert(0, ".")
from diff3dhpe_amd.engine import op_linear, op_attention
torch.manual_seed(0)
bf = lambda x: x.to(torch.bfloat16).to(torch.float64)
M, N, K = 1024, 512, 512
A = torch.randn(M, K); W = torch.randn(N, K) / K ** 0.5; b = torch.randn(N)
for epi in ("none", "gelu"):
    out = op_linear(A.cuda(), W.cuda(), b.cuda(), None, epi=epi, precision="bf16").cpu().double()
    ref = bf(A) @ bf(W).T + b.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    rne = bf(ref.float())
    trunc = (ref.float().view(torch.int32) & ~0xFFFF).view(torch.float32).double()
    print(f"linear {epi}: equals RNE {(out == rne).float().mean():.5f}  equals truncation {(out == trunc).float().mean():.5f}  mean signed err/|ref| {((out - ref) / ref.abs().clamp_min(1e-3)).mean():.3e}")
B, T, J, D, H = 2, 81, 17, 512, 8
qkv = torch.randn(B * T * J, 3 * D)
out = op_attention(qkv.cuda(), B, T, J, H, True, precision="bf16").cpu().double()
x = qkv.view(B, T, J, 3, H, 64).permute(3, 0, 2, 4, 1, 5)
q, k, v = bf(x[0]), bf(x[1]), bf(x[2])
p = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1) - torch.eye(T, dtype=torch.float64)
for name, pp in (("P rounded RNE", bf(p.float())), ("P not rounded", p), ("P truncated", (p.float().view(torch.int32) & ~0xFFFF).view(torch.float32).double())):
    o = (pp @ v).permute(0, 3, 1, 2, 4).reshape(B * T * J, D)
    print(f"attention, reference with {name}: out equals RNE(ref) {(out == bf(o.float())).float().mean():.5f}  max-abs {(out - o).abs().max():.3e} rms {(out - o).pow(2).mean().sqrt():.3e}")
