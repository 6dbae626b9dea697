"""Per-kernel parity on the MI355X, through the C ABI (d3d_op_*): each HIP kernel against fp64 math of the same op
and against the reference-derived golden vectors.  Tolerances are absolute fp32-rounding bounds for O(1) data."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import gold
from helpers import hashed, torch_sd, maxabs
from diff3dhpe_amd.spec import DenoiserConfig

pytestmark = pytest.mark.gpu


def _eng():
    from diff3dhpe_amd import engine
    return engine


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 96, 32), (1000, 1536, 512), (2066, 512, 1024), (17, 64, 64),
                                   (4131, 1024, 512), (129, 130, 96)])
def test_linear_matches_fp64(M, N, K, prec):
    """Both GEMM paths must sit at fp32-rounding distance from the fp64 product (F16X3 = 3 fp16 MFMAs on hi/lo splits)."""
    E = _eng()
    A = hashed(f"A{M}", (M, K), 11, 2.0).cuda()
    W = hashed(f"W{N}", (N, K), 12, 1.0 / np.sqrt(K)).cuda()
    b = hashed(f"b{N}", (N,), 13, 0.5).cuda()
    R = hashed(f"R{M}", (M, N), 14, 1.0).cuda()
    ref = A.double() @ W.double().t() + b.double()
    tol = 2e-6 * np.sqrt(K / 32)
    assert maxabs(E.op_linear(A, W, b, precision=prec), ref.cpu()) < tol
    assert maxabs(E.op_linear(A, W, None, precision=prec), (ref - b.double()).cpu()) < tol
    assert maxabs(E.op_linear(A, W, b, epi="gelu", precision=prec), F.gelu(ref).cpu()) < tol
    assert maxabs(E.op_linear(A, W, b, residual=R, epi="residual", precision=prec), (ref + R.double()).cpu()) < tol


def test_linear_f16x3_dynamic_range():
    """hi/lo split must survive small and large activations / weights (scales 2^3 and 2^12 keep lo a normal fp16)."""
    E = _eng()
    for a_scale, w_scale in ((1e-3, 0.05), (30.0, 0.05), (1.0, 1e-3), (200.0, 0.5)):
        A = hashed("Adyn", (256, 512), 5, a_scale).cuda()
        W = hashed("Wdyn", (256, 512), 6, w_scale).cuda()
        ref = A.double() @ W.double().t()
        got = E.op_linear(A, W, None, precision="f16x3")
        scale = ref.abs().max().item()
        # relative fp32-level accuracy, plus the absolute floor of a (sub)normal fp16 lo part (2^-24/8 per operand)
        assert maxabs(got, ref.cpu()) < 3e-6 * scale + 2e-8, (a_scale, w_scale)


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_linear_identity_with_asymmetric_weight(prec):
    """A = I against an asymmetric W catches a transposed C/D map (cdna guide section 3)."""
    E = _eng()
    K = 64
    A = torch.eye(K, device="cuda")
    W = ((torch.arange(96 * K, dtype=torch.float32).reshape(96, K) % 13) * 0.25).cuda()
    out = E.op_linear(A, W, None, precision=prec)
    assert torch.equal(out.cpu(), W.t().cpu())


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_linear_bit_reproducible_and_row_independent(prec):
    E = _eng()
    A = hashed("Arep", (777, 512), 1, 2.0).cuda()
    W = hashed("Wrep", (512, 512), 2, 0.05).cuda()
    o1, o2 = E.op_linear(A, W, None, precision=prec), E.op_linear(A, W, None, precision=prec)
    assert torch.equal(o1, o2)
    assert torch.equal(E.op_linear(A[100:229].contiguous(), W, None, precision=prec), o1[100:229])   # tile position must not matter
    # ... nor the epilogue form an element goes through: M = 66100 runs the persistent walk (whole tiles: 8 columns per lane; the
    # ragged last tile and the row slices: 4 columns per lane), the 129-row excerpt the small-problem tiles -- with GELU (one
    # implementation behind every form)
    Ab = hashed("Arep2", (66100, 512), 3, 2.0).cuda()
    bias = hashed("brep", (512,), 4, 0.5).cuda()
    big = E.op_linear(Ab, W, bias, epi="gelu", precision=prec)
    for lo in (0, 65900, 65971):
        assert torch.equal(E.op_linear(Ab[lo:lo + 129].contiguous(), W, bias, epi="gelu", precision=prec), big[lo:lo + 129]), lo


@pytest.mark.parametrize("M,K,mode", [(128, 2048, "plain"), (300, 512, "pos"), (4131, 2048, "tvec1"), (1000, 64, "tvecrows"),
                                      (17, 32, "all"), (140000, 512, "big")])
def test_linear_postnorm_matches_fp64(M, K, mode):
    """fc2 + post-norm in one GEMM (whole-row 128x512 tiles, LayerNorm in the epilogue; S2S:131-135 + 236/245 with the
    additions of S2S:238-242 / 113-116): both output forms against fp64 math, the row statistics handed to the next folded
    GEMM, and bitwise independence of a row from the tile position / launch form (M = 140000 takes the persistent walk)."""
    E = _eng()
    N = 512
    A = hashed(f"pnA{M}", (M, K), 21, 2.0).cuda()
    W = hashed(f"pnW{K}", (N, K), 22, 1.0 / np.sqrt(K)).cuda()
    b = hashed("pnb", (N,), 23, 0.5).cuda()
    R = (hashed(f"pnR{M}", (M, N), 24, 1.5) + 0.3).cuda()
    g = (1 + 0.2 * hashed("png", (N,), 25)).cuda()
    be = (0.2 * hashed("pnbe", (N,), 26)).cuda()
    rows = torch.arange(M, device="cuda")
    kw, add = {}, torch.zeros((M, N), dtype=torch.float64, device="cuda")
    if mode in ("pos", "all"):
        pos = hashed("pnpos", (9, N), 27, 0.5).cuda()
        kw.update(pos=pos, pos_div=17)
        add += pos.double()[(rows // 17) % 9]
    if mode in ("tvec1", "big"):
        tv = hashed("pntv", (N,), 28, 0.5).cuda()
        kw.update(tvec=tv)
        add += tv.double()
    if mode in ("tvecrows", "all"):
        rpb = 51
        tv = hashed("pntvr", ((M + rpb - 1) // rpb, N), 29, 0.5).cuda()
        kw.update(tvec=tv, rows_per_batch=rpb)
        if tv.shape[0] > 1:
            add += tv.double()[rows // rpb]
        else:
            add += tv.double()[0]
    ref = F.layer_norm(R.double() + A.double() @ W.double().t() + b.double(), (N,), g.double(), be.double(), 1e-6) + add
    tol = 3e-6 * np.sqrt(K / 32) + 2e-6
    y32, _, _ = E.op_linear_postnorm(A, W, b, R, g, be, 1e-6, **kw)
    assert maxabs(y32, ref.cpu()) < tol
    yp, st, _ = E.op_linear_postnorm(A, W, b, R, g, be, 1e-6, with_stats=True, **kw)
    assert maxabs(yp, ref.cpu()) < tol + 2e-6                     # planes hold 22 bits of y
    assert maxabs(yp, y32.cpu()) < 2e-6
    assert maxabs(st[:, 0], ref.sum(1).cpu()) < 2e-3 and maxabs(st[:, 1], (ref * ref).sum(1).cpu()) < 1e-2
    lo, hi = (100, 229) if M > 229 else (0, M)
    part, _, _ = E.op_linear_postnorm(A[lo:hi].contiguous(), W, b, R[lo:hi].contiguous(), g, be, 1e-6)
    if not kw:
        assert torch.equal(part, y32[lo:hi])
    else:   # (row classes shift with the slice: compare the normalised part only)
        part0, _, _ = E.op_linear_postnorm(A[lo:hi].contiguous(), W, b, R[lo:hi].contiguous(), g, be, 1e-6)
        full0, _, _ = E.op_linear_postnorm(A, W, b, R, g, be, 1e-6)
        assert torch.equal(part0, full0[lo:hi])


def test_linear_postnorm_rejects_other_widths():
    E = _eng()
    from diff3dhpe_amd import D3DError
    A, W, R = torch.zeros(64, 64).cuda(), torch.zeros(256, 64).cuda(), torch.zeros(64, 256).cuda()
    v = torch.zeros(256).cuda()
    with pytest.raises(D3DError):
        E.op_linear_postnorm(A, W, v, R, v, v)


@pytest.mark.parametrize("rows,D,eps", [(5, 32, 1e-6), (1000, 512, 1e-6), (333, 512, 1e-5), (64, 1024, 1e-6), (3, 128, 1e-6)])
def test_layernorm(rows, D, eps):
    E = _eng()
    x = hashed("lnx", (rows, D), 3, 3.0).cuda() + 0.7
    g = (1 + 0.1 * hashed("lng", (D,), 4)).cuda()
    b = (0.1 * hashed("lnb", (D,), 5)).cuda()
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), eps)
    assert maxabs(E.op_layernorm(x, g, b, eps), ref.cpu()) < 3e-6


def _attn_ref(qkv, B, T, J, H, temporal):
    D = qkv.shape[-1] // 3
    dh = D // H
    x = qkv.double().reshape(B, T, J, 3, H, dh)
    if temporal:
        x = x.permute(0, 2, 1, 3, 4, 5)            # (B, J, T, 3, H, dh): groups are joints
    q, k, v = (x[..., i, :, :].transpose(-3, -2) for i in range(3))   # (.., H, N, dh)
    a = (q @ k.transpose(-2, -1)) * dh ** -0.5
    a = a.softmax(-1)
    N = a.shape[-1]
    o = (a - torch.eye(N, dtype=a.dtype, device=a.device)) @ v          # (B, G2, H, N, dh)
    o = o.transpose(-3, -2)                                             # (B, G2, N, H, dh)
    if temporal:
        o = o.permute(0, 2, 1, 3, 4)                                    # (B, T, J, H, dh)
    return o.reshape(B * T * J, D)


@pytest.mark.parametrize("B,T,J,D,H,temporal,generic", [
    (2, 5, 17, 512, 8, False, False), (2, 5, 17, 512, 8, False, True), (3, 81, 17, 512, 8, False, False),
    (1, 27, 17, 512, 8, True, False), (2, 81, 3, 512, 8, True, False), (1, 243, 2, 512, 8, True, False),
    (1, 243, 2, 512, 8, True, True), (1, 256, 1, 512, 8, True, False), (1, 33, 2, 128, 2, True, False),
    (2, 9, 17, 32, 8, False, False), (2, 27, 17, 32, 8, True, False), (1, 100, 2, 64, 8, True, False),
    (1, 160, 1, 512, 8, True, False), (1, 1, 17, 512, 8, True, False),
])
def test_attention_core(B, T, J, D, H, temporal, generic):
    E = _eng()
    qkv = hashed(f"qkv{T}_{J}_{D}", (B * T * J, 3 * D), 21, 2.0).cuda()
    out = E.op_attention(qkv, B, T, J, H, temporal, force_generic=generic)
    ref = _attn_ref(qkv, B, T, J, H, temporal)
    assert maxabs(out, ref.cpu()) < 5e-6
    if temporal and not generic:   # the fp16-MFMA (F16X3) temporal kernel: planes in, planes out, fp32-level accuracy
        out3 = E.op_attention(qkv, B, T, J, H, True, precision="f16x3")
        assert maxabs(out3, ref.cpu()) < 5e-6


@pytest.mark.parametrize("B,T,J,temporal", [(8, 243, 17, True), (8, 256, 17, True), (8, 230, 17, True), (24, 81, 17, True),
                                            (3, 243, 17, False), (9, 81, 17, False)])
def test_attention_persistent_kernels(B, T, J, temporal):
    """The forms only large launches take -- the staggered persistent temporal kernel (k_attn_temporal_x3s: >= 1024 units and 8
    key tiles, here with 13, 0 and 26 pad rows in the last tile), the persistent 3-tile kernel (T = 81) and the wave-private
    persistent spatial kernel (>= 4096 units) -- against fp64 math on a sample of rows, and bit for bit against the same rows
    computed as a small launch (the non-persistent kernels the other tests cover)."""
    E = _eng()
    D, H = 512, 8
    qkv = hashed(f"pers{T}_{B}_{int(temporal)}", (B * T * J, 3 * D), 23, 2.0).cuda()
    if not temporal:     # the engine runs spatial blocks as groups of J consecutive tokens: B*T groups of length J, stride 1
        B, T, J = B * T, J, 1
    out = E.op_attention(qkv, B, T, J, H, True, precision="f16x3")
    n1 = (1 if T > 32 else 8) * T * J                       # a launch small enough for the non-persistent kernels
    b1 = n1 // (T * J)
    one = slice(0, n1)
    ref = _attn_ref(qkv[one], b1, T, J, H, True)
    assert maxabs(out[one], ref.cpu()) < 5e-6
    small = E.op_attention(qkv[one].contiguous(), b1, T, J, H, True, precision="f16x3")
    assert torch.equal(out[one], small)
    last = slice(B * T * J - n1, B * T * J)
    assert torch.equal(out[last], E.op_attention(qkv[last].contiguous(), b1, T, J, H, True, precision="f16x3"))


def test_attention_fast_kernels_agree_with_generic_on_sharp_softmax():
    """Large logits (|s| up to ~100, near one-hot softmax) exercise the max-subtraction and the -inf key mask.
    fp32 logits carry ~3e-5 absolute error at this scale (64-term dots of products up to 81), which exp() turns into a
    3e-5 RELATIVE error of the weights; outputs are O(10), so the fp32 bound is ~1e-3 -- torch's own fp32 path sits
    at the same distance from fp64."""
    E = _eng()
    for (B, T, J, temporal) in ((2, 7, 17, False), (1, 243, 3, True), (1, 81, 2, True)):
        qkv = hashed(f"sharp{T}", (B * T * J, 3 * 512), 22, 9.0).cuda()
        a = E.op_attention(qkv, B, T, J, 8, temporal)
        b = E.op_attention(qkv, B, T, J, 8, temporal, force_generic=True)
        ref = _attn_ref(qkv, B, T, J, 8, temporal)
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        assert maxabs(a, ref.cpu()) < 2e-3 and maxabs(b, ref.cpu()) < 2e-3
        assert maxabs(a, b.cpu()) < 2e-3
        if temporal:
            c = E.op_attention(qkv, B, T, J, 8, True, precision="f16x3")
            assert torch.isfinite(c).all() and maxabs(c, ref.cpu()) < 2e-3


GOLD_ATTN = [("spatial_D512", 512, 17, 2), ("spatial_D32", 32, 17, 4), ("temporal_D512_T27", 512, 27, 1),
             ("temporal_D512_T81", 512, 81, 1), ("temporal_D512_T243", 512, 243, 1), ("temporal_D32_T81", 32, 81, 2)]


@pytest.mark.parametrize("tag,D,N,G", GOLD_ATTN, ids=[g[0] for g in GOLD_ATTN])
def test_attention_module_golden(tag, D, N, G):
    """Attention.forward (S2S:73-86) = qkv GEMM -> GRAND core -> proj GEMM against the reference's own output."""
    E = _eng()
    g = gold("attention")
    sd = {k: v.cuda() for k, v in torch_sd(DenoiserConfig(num_frame=9, embed_dim=D, depth=1), 2).items()}
    x = hashed("attn_in/" + tag, (G, N, D), 2, 1.5).cuda()
    spatial = tag.startswith("spatial")
    p = ("STEblocks.0" if spatial else "TTEblocks.0") + ".attn"
    qkv = E.op_linear(x.reshape(G * N, D), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    if spatial:      # G frames of N=17 joints
        core = E.op_attention(qkv, 1, G, N, 8, False)
    else:            # G joints-groups of N frames: token order (b=G? no: B=G, T=N, J=1)
        core = E.op_attention(qkv, G, N, 1, 8, True)
    out = E.op_linear(core, sd[p + ".proj.weight"], sd[p + ".proj.bias"]).reshape(G, N, D)
    assert maxabs(out, g[tag]) < 1e-5


BLOCKS = [("ste_D512", 512, (1, 4, 17, 512)), ("tte_D512_T81", 512, (1, 81, 2, 512)), ("ste_D32", 32, (2, 9, 17, 32)),
          ("tte_D32_T27", 32, (2, 27, 17, 32))]


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag,D,shape", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_block_golden(tag, D, shape, prec):
    """One Block.forward (S2S:111-135) and the post-norm behind it (S2S:236 / 245) at block granularity, composed from the
    single-op hooks (LayerNorm, qkv GEMM, GRAND core, proj GEMM + residual, LayerNorm, fc1 GEMM + GELU, fc2 GEMM + residual,
    post-norm) in token-major layout, against the reference's own block output (tests/golden/blocks.npz): a wrong-but-
    compensating pair of kernels inside a block cannot hide behind the end-to-end gate."""
    E = _eng()
    g = gold("blocks")
    b, f, j, _ = shape
    sd = {k: v.cuda() for k, v in torch_sd(DenoiserConfig(num_frame=f, embed_dim=D, depth=1), 3).items()}
    x = hashed("block_in/" + tag, shape, 3, 1.2).cuda()
    temb = hashed("block_temb/" + tag, (b, 2 * D), 3).cuda()
    sp = tag.startswith("ste")
    p = "STEblocks.0" if sp else "TTEblocks.0"
    # the per-block time vector (S2S:113-116) is host-side test preparation here; its kernel has its own test (time-embedding table)
    te = F.linear(F.silu(temb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    x1 = (x + te[:, None, None, :]).reshape(b * f * j, D).contiguous()
    lin = lambda a, w, bias, **kw: E.op_linear(a, sd[p + w], sd[p + bias], precision=prec, **kw)
    h = E.op_layernorm(x1, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
    qkv = lin(h, ".attn.qkv.weight", ".attn.qkv.bias")
    core = E.op_attention(qkv, b, f, j, 8, not sp, precision=prec)
    x2 = lin(core, ".attn.proj.weight", ".attn.proj.bias", residual=x1, epi="residual")
    h2 = E.op_layernorm(x2, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    hid = lin(h2, ".mlp.fc1.weight", ".mlp.fc1.bias", epi="gelu")
    x3 = lin(hid, ".mlp.fc2.weight", ".mlp.fc2.bias", residual=x2, epi="residual")
    post = "Spatial_norm" if sp else "Temporal_norm"
    z = E.op_layernorm(x3, sd[post + ".weight"], sd[post + ".bias"], 1e-6)
    assert maxabs(x3.reshape(shape), g[tag + "/block"]) < 2e-5, (tag, prec)
    assert maxabs(z.reshape(shape), g[tag + "/postnorm"]) < 2e-5, (tag, prec)
    if prec == "f16x3" and D == 512:      # the fused form the engine runs: fc2 + post-norm in one GEMM epilogue
        y, _, _ = E.op_linear_postnorm(hid, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"], x2, sd[post + ".weight"], sd[post + ".bias"], 1e-6)
        assert maxabs(y.reshape(shape), g[tag + "/postnorm"]) < 2e-5, tag


@pytest.mark.parametrize("rows", [1, 33, 4131, 40000])
def test_head_kernel(rows):
    """k_head (S2S:217-220: LayerNorm eps 1e-5 + Linear D -> 3) on its own, through d3d_op_head: against fp64 math, for row counts
    below, at and far above the 32 rows a workgroup owns, and twice in a row bit for bit."""
    from diff3dhpe_amd.engine import Engine
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=2)
    sd = torch_sd(cfg, 11)
    eng = Engine(cfg, precision="f16x3")
    eng.load_weights(sd)
    X = hashed(f"head{rows}", (rows, 512), 12, 1.7).cuda() + 0.3
    out = eng.head(X)
    g, b = sd["head.0.weight"].double().cuda(), sd["head.0.bias"].double().cuda()
    W, c = sd["head.1.weight"].double().cuda(), sd["head.1.bias"].double().cuda()
    xd = X.double()
    ref = ((xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * g + b) @ W.t() + c
    assert maxabs(out, ref.cpu()) < 3e-6
    assert torch.equal(out, eng.head(X))
