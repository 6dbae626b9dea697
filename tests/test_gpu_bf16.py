"""bf16 operand mode (D3D_PREC_BF16; BASELINE configs[1], SURVEY section 7 step 5 / section 8(d) "Parity gates") on the MI355X.

A second-class precision: bf16 operands cannot meet the 1e-4 gate (~5e-2 max-abs at random init), so the
engine is gated against the CPU oracle's *bf16-operand emulation* (oracle.operand_rounding: the operands of the four block GEMMs
and of both attention products rounded to bf16, everything else fp32) at the bound SURVEY section 8(d) gives -- <= 2e-3
normalised MPJPE and <= 2e-2 max-abs -- and its distance to the fp32 oracle is REPORTED (printed), not gated."""
import numpy as np
import pytest
import torch

from helpers import cfg_full, inputs, build_product, maxabs, torch_sd
from diff3dhpe_amd.engine import op_linear, op_attention
from diff3dhpe_amd.spec import DenoiserConfig

pytestmark = pytest.mark.gpu
GATE_MAXABS, GATE_MPJPE = 2e-2, 2e-3


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _mpjpe(a, b):
    return (a.detach().cpu().double() - b.double()).norm(dim=-1).mean().item()


@pytest.mark.parametrize("M,N,K,epi", [(300, 512, 512, "none"), (1000, 1536, 512, "none"), (517, 1024, 512, "gelu"),
                                        (517, 512, 1024, "residual"), (70000, 512, 512, "residual"), (70000, 1024, 512, "gelu")])
def test_bf16_linear_matches_rounded_operand_math(M, N, K, epi):
    """One bf16 MFMA per product, fp32 accumulation: against fp64 math on the SAME bf16-rounded operands the only differences are
    the fp32 accumulation order and, for the bf16-output forms, the final rounding (<= 1 bf16 ulp).  70 000 rows run the 256x256
    persistent walk with tail slices, the small cases the 256x128 tiles with ragged edges."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if epi == "residual" else None
    out = op_linear(A.cuda(), W.cuda(), b.cuda(), R.cuda() if R is not None else None, epi=epi, precision="bf16").cpu().double()
    ref = _bf(A) @ _bf(W).T + b.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "residual":
        ref = ref + R.double()
        assert (out - ref).abs().max().item() <= 2e-5 * (K / 512) ** 0.5 * 8        # fp32 accumulation only
    else:
        ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -120, dtype=torch.float64)) * 2.0 ** -8
        assert ((out - ref).abs() <= ulp + 2e-5).all()                               # one bf16 rounding of the output
        assert torch.equal(out, _bf(out.float()))                                    # the output IS bf16-valued


@pytest.mark.parametrize("B,T,J,temporal", [(2, 243, 17, True), (3, 81, 5, True), (2, 27, 17, True), (2, 9, 17, False), (70, 27, 17, False)])
def test_bf16_attention_matches_rounded_operand_math(B, T, J, temporal):
    """(softmax(q k^T / 8) - I) v with q, k, v and softmax - I rounded to bf16: the kernel against fp64 math with the same roundings
    (a flipped rounding of one P element costs <= 2^-9 |v|, so the bound is a few bf16 ulps of the output scale)."""
    D, H = 512, 8
    g = torch.Generator().manual_seed(B * T)
    qkv = torch.randn(B * T * J, 3 * D, generator=g)
    out = op_attention(qkv.cuda(), B, T, J, H, temporal, precision="bf16").cpu().double()
    x = qkv.view(B, T, J, 3, H, 64)
    if temporal:
        x = x.permute(3, 0, 2, 4, 1, 5)          # (3, B, J, H, T, dh)
    else:
        x = x.permute(3, 0, 1, 4, 2, 5)          # (3, B, T, H, J, dh)
    q, k, v = _bf(x[0]), _bf(x[1]), _bf(x[2])
    a = (q @ k.transpose(-2, -1)) * 0.125
    p = a.softmax(-1) - torch.eye(a.shape[-1], dtype=torch.float64)
    o = _bf(_bf(p.float()) @ v)                   # the kernel writes bf16
    o = o.permute(0, 3, 1, 2, 4) if temporal else o.permute(0, 1, 3, 2, 4)     # (B, T, J, H, dh)
    ref = o.reshape(B * T * J, D)
    err = (out - ref).abs().max().item()
    print(f"bf16 attention B={B} T={T} J={J} temporal={temporal}: max-abs vs rounded-operand fp64 {err:.3e} (|out| max {ref.abs().max():.2f})")
    assert err <= 3e-2 * max(1.0, ref.abs().max().item() / 4)


CASES = [("T27", cfg_full(27), 2, 9), ("T81", cfg_full(81), 2, 5), ("T243", cfg_full(243), 1, 3),
         ("s2f_T27", cfg_full(27, seq2frame=True), 2, 5), ("notemb_T27", cfg_full(27, with_time_emb=False), 2, 5)]


@pytest.mark.parametrize("tag,cfg,B,S", CASES, ids=[c[0] for c in CASES])
def test_bf16_engine_against_the_oracles_bf16_emulation(tag, cfg, B, S):
    from oracle import d3d_oracle as orc
    net, diff = build_product(cfg, 91, sampling=S, precision="bf16")
    inp = inputs(B, cfg.num_frame, 910)
    sd = torch_sd(cfg, 91)
    tabs = orc.diffusion_tables("cosine", 1000)
    # one denoiser evaluation (raw, unclamped output), per-row timesteps
    xcat = torch.cat([inp["x2d"], inp["noise"] * 0.7], dim=-1)
    t = torch.tensor([(431 * i + 77) % 1000 for i in range(B)], dtype=torch.long)
    out = net.forward_denoise(xcat.cuda(), t.cuda())
    with orc.operand_rounding(torch.bfloat16):
        emu = orc.forward_denoise(sd, xcat, t, depth=cfg.depth, seq2frame=cfg.seq2frame)
    f32 = orc.forward_denoise(sd, xcat, t, depth=cfg.depth, seq2frame=cfg.seq2frame)
    e1, m1 = maxabs(out, emu), _mpjpe(out, emu)
    print(f"bf16 denoise {tag}: vs bf16 emulation max-abs {e1:.3e} MPJPE {m1:.3e} | vs fp32 oracle max-abs {maxabs(out, f32):.3e} "
          f"MPJPE {_mpjpe(out, f32):.3e} | emulation vs fp32 oracle max-abs {(emu - f32).abs().max():.3e}")
    assert e1 <= GATE_MAXABS and m1 <= GATE_MPJPE
    # the whole sampling
    noise = inp["noise"][:, :1].contiguous() if cfg.seq2frame else inp["noise"]
    _, y0 = diff(clean_3d_pose=torch.zeros_like(noise).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False, init_noise=noise.cuda())
    kw = dict(num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth, seq2frame=cfg.seq2frame)
    with orc.operand_rounding(torch.bfloat16):
        emu = orc.ddim_sample_loop(sd, tabs, inp["x2d"], noise, **kw)
    f32 = orc.ddim_sample_loop(sd, tabs, inp["x2d"], noise, **kw)
    e2, m2 = maxabs(y0, emu), _mpjpe(y0, emu)
    print(f"bf16 ddim {tag} S={S}: vs bf16 emulation max-abs {e2:.3e} MPJPE {m2:.3e} | vs fp32 oracle max-abs {maxabs(y0, f32):.3e} "
          f"MPJPE {_mpjpe(y0, f32):.3e} ({_mpjpe(y0, f32) * 1000:.2f} 'mm at scale 1.0')")
    assert e2 <= GATE_MAXABS and m2 <= GATE_MPJPE
    assert y0.abs().max().item() <= 1.0 and torch.isfinite(y0).all()


def test_bf16_large_batch_kernels_match_the_small_batch_path():
    """B = 32 at T = 243 runs the persistent 256x256 bf16 GEMM walk (tail slices) and eight-wave attention workgroups; every output
    element is tile-shape independent, so the large batch reproduces ragged small chunks bit for bit, twice; two streams too."""
    cfg = cfg_full(243)
    _, diff = build_product(cfg, 8, sampling=1, precision="bf16")
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    inp = inputs(32, 243, 5)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    big = eng.ddim_sample(x2d, nz).clone()
    assert torch.equal(big, eng.ddim_sample(x2d, nz))
    for lo in (0, 13, 26):
        hi = min(lo + 13, 32)
        assert torch.equal(eng.ddim_sample(x2d[lo:hi].contiguous(), nz[lo:hi].contiguous()), big[lo:hi])
    eng.set_option("streams", 2)
    assert torch.equal(eng.ddim_sample(x2d, nz), big)
    eng.set_option("streams", 1)
    assert torch.isfinite(big).all() and big.abs().max().item() <= 1.0


def test_bf16_mode_refuses_shapes_it_has_no_kernels_for():
    import diff3dhpe_amd as d3d
    from diff3dhpe_amd import _lib
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=9, embed_dim=32, depth=1)       # head_dim 4
    net.precision = "bf16"
    with pytest.raises(_lib.D3DError, match="BF16"):
        net.engine_for(torch.device("cuda", torch.cuda.current_device()))
