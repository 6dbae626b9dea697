"""The fc2 + post-norm GEMM alone (d3d_op_linear_postnorm), token GEMM template vs the ring kernel: which tiles differ, ms per launch.
   python experiments/fc2_ring_op.py [M delay dbg reps]"""
import ctypes as C
import sys
import torch
sys.path[:0] = ["."]
from diff3dhpe_amd import engine as E, _lib

M, delay, dbg, reps, K = (int(x) for x in (sys.argv[1:6] + ["264384", "24", "0", "20", "1024"][len(sys.argv) - 1:]))
N = 512
g = torch.Generator().manual_seed(3)
A = torch.randn(M, K, generator=g).cuda() * 0.5
W = (torch.rand(N, K, generator=g).cuda() - 0.5) * (2.0 / 32.0)
b = torch.randn(N, generator=g).cuda() * 0.1
R = torch.randn(M, N, generator=g).cuda()
ga = 1.0 + 0.1 * torch.randn(N, generator=g).cuda()
be = 0.1 * torch.randn(N, generator=g).cuda()
opt = lambda k, v: _lib.check(_lib.lib().d3d_engine_set_option(None, k.encode(), C.c_int64(v)))
opt("fc2_ring_op", 0)
ref, _, ms0 = E.op_linear_postnorm(A, W, b, R, ga, be, 1e-6, reps=reps)
opt("fc2_ring_op", 1); opt("fc2_ring_delay", delay); opt("fc2_ring_dbg", dbg)
for r in range(3):
    out, _, ms1 = E.op_linear_postnorm(A, W, b, R, ga, be, 1e-6, reps=reps)
    bad = (out != ref).any(dim=1)
    tiles = torch.unique(bad.nonzero().flatten() // 128)
    d = (out - ref).abs().max().item()
    info = [(int(t) % 256, int(t) // 256) for t in tiles[:12]]
    print(f"M {M} K {K} delay {delay} dbg {dbg}: template {ms0:.4f} ms, ring {ms1:.4f} ms | bad rows {int(bad.sum())} tiles {tiles.numel()} max diff {d:.3e} (wg, item) {info}")
    if tiles.numel():
        t = int(tiles[0]); rows = bad[t * 128:(t + 1) * 128].nonzero().flatten().tolist()
        cols = (out[t * 128 + rows[0]] != ref[t * 128 + rows[0]]).nonzero().flatten().tolist()
        print("   ref", ref[t * 128 + rows[0], cols[:4]].tolist(), "out", out[t * 128 + rows[0], cols[:4]].tolist(), "ref next col", ref[t * 128 + rows[0], [c + 1 for c in cols[:4]]].tolist())
        r0 = t * 128 + rows[0]; blk = cols[0] // 64
        pre = (R[r0].double() + A[r0].double() @ W.double().T + b.double())
        mean, var = pre.mean(), pre.var(unbiased=False)
        lm = pre[blk * 64:(blk + 1) * 64].mean()
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        print("   observed diff", (out[r0, cols[:4]] - ref[r0, cols[:4]]).tolist(), "predicted -lm rstd gamma", (-lm * rstd * ga[cols[:4]].double()).tolist(), "lm", float(lm), "mean", float(mean))
        print(f"   tile {t}: bad rows in tile {rows[:20]}{'...' if len(rows) > 20 else ''} ({len(rows)}); bad cols of first: {cols[:16]} ({len(cols)})")

# in-kernel stamps (wave 0 and wave 4 of every workgroup): k-loop / epilogue cycles per tile, counter polls per tile
dg = torch.zeros(256 * 22, dtype=torch.int64, device="cuda")
opt("fc2_ring_diag", dg.data_ptr())
out, _, ms2 = E.op_linear_postnorm(A, W, b, R, ga, be, 1e-6, reps=1)
opt("fc2_ring_diag", 0)
d = dg.cpu()[:256 * 8].reshape(256, 2, 4).double()
sg = dg.cpu()[256 * 8:].reshape(256, 2, 7).double()
for h, name in ((0, "wave 0"), (1, "wave 4")):
    tiles = d[:, h, 2]
    print(f"   {name}: k-loop {float((d[:, h, 0] / tiles).median()):.0f} cycles per tile = {float((d[:, h, 0] / tiles / (K // 32)).median()):.0f} per k-tile, "
          f"epilogue {float((d[:, h, 1] / tiles).median()):.0f}, polls per k-tile {float((d[:, h, 3] / tiles / (K // 32)).median()):.2f} (max {float((d[:, h, 3] / tiles / (K // 32)).max()):.2f})")
    if sg.abs().sum() > 0:   # -DR2_STAMPS library: per k-tile segments
        kt = tiles * (K // 32)
        names = ["wait own pieces (vmcnt)", "W frags + poll A slot", "A frags + group 0", "groups 1-3 + signal", "groups 4-7"]
        print("      " + "; ".join(f"{n} {float((sg[:, h, i] / kt).median()):.0f}" for i, n in enumerate(names[:5])) +
              f"; epilogue sweep 1 {float((sg[:, h, 6] / tiles).median()):.0f} of {float((d[:, h, 1] / tiles).median()):.0f} cycles per tile")
