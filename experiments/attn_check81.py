import sys, torch
sys.path.insert(0, "/root/repo")
from diff3dhpe_amd.engine import op_attention
B, T, J, D, H = 24, 81, 17, 512, 8
g = torch.Generator().manual_seed(3)
qkv = torch.randn(B * T * J, 3 * D, generator=g).cuda()
big = op_attention(qkv, B, T, J, H, True, "f16x3")
small = op_attention(qkv[:T * J].contiguous(), 1, T, J, H, True, "f16x3")
d = big[:T * J] - small
nz = (d != 0).nonzero()
print("mismatching", nz.shape[0], "of", d.numel(), "max", d.abs().max().item())
if nz.shape[0]:
    rows = nz[:, 0]; t = rows // J; cols = nz[:, 1]
    print("t hist (t%32)", torch.bincount(t % 32, minlength=32).tolist())
    print("t//32", torch.bincount(t // 32, minlength=3).tolist())
    print("col%64 hist", torch.bincount(cols % 64, minlength=64).tolist())
    print("head hist", torch.bincount(cols // 64, minlength=8).tolist())
    # is big a permutation of small within rows?
    r0 = rows[0].item()
    print("row", r0, "big", big[r0, :16].tolist(), "small", small[r0, :16].tolist())
# which small-launch row does each wrong big-launch row hold?  (joint 0, head 0, first line = columns 0..31)
sm = small.reshape(T, J, D)[:, 0, :]
bg = big[:T * J].reshape(T, J, D)[:, 0, :]
for t in (8, 9, 10, 11, 12, 16, 17, 24, 25, 40):
    for lo, name in ((0, "line0"), (32, "line1")):
        d = (sm[:, lo:lo + 32][None, :, :] - bg[t, lo:lo + 32][None, None, :]).abs().amax(-1)[0]
        best = int(d.argmin())
        # per 8-column chunk
        chunks = []
        for c in range(4):
            dc = (sm[:, lo + 8 * c:lo + 8 * c + 8] - bg[t, lo + 8 * c:lo + 8 * c + 8][None, :]).abs().amax(-1)
            chunks.append((int(dc.argmin()), float(dc.min())))
        print("t", t, name, "best row", best, "err", float(d.min()), "per-chunk source rows", chunks)
