import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from diff3dhpe_amd.engine import op_attention
B, T, J, D, H = 16, 243, 17, 512, 8
g = torch.Generator().manual_seed(3)
qkv = torch.randn(B * T * J, 3 * D, generator=g).cuda()
big = op_attention(qkv, B, T, J, H, True, "f16x3")
bad = 0
for b in range(0, B, 4):
    small = op_attention(qkv[b * T * J:(b + 4) * T * J].contiguous(), 4, T, J, H, True, "f16x3")
    d = (big[b * T * J:(b + 4) * T * J] - small)
    nz = (d != 0).nonzero()
    print(b, "mismatching elements", nz.shape[0], "max", d.abs().max().item())
    if nz.shape[0]:
        rows = nz[:, 0].unique()
        t = (rows // J) % T; j = rows % J; bb = rows // (T * J)
        print("   rows", rows.shape[0], "t range", t.min().item(), t.max().item(), "t mod 32 hist", torch.bincount(t % 32, minlength=32).tolist())
        print("   cols (head)", torch.bincount(nz[:, 1] // 64, minlength=8).tolist(), "col in head hist", torch.bincount(nz[:, 1] % 64, minlength=64).tolist())
        print("   t//32 hist", torch.bincount(t // 32, minlength=8).tolist())
