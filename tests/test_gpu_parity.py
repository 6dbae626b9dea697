"""End-to-end parity of the HIP engine on the MI355X against the golden vectors captured from the reference
(tests/golden, oracle/gen_golden.py) -- through the product's reference-shaped API (HPE_model / GaussianDiffusion),
i.e. through the C ABI.  Gate (BASELINE.json north_star): max-abs <= 1e-4 in fp32; DDIM index schedule bit-exact."""
import numpy as np
import pytest
import torch

from conftest import gold
from helpers import cfg_small, cfg_full, inputs, hashed, build_product, maxabs, torch_sd
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig

pytestmark = pytest.mark.gpu
GATE = 1e-4


def test_library_is_loaded_and_device_is_mi355x():
    import diff3dhpe_amd
    from diff3dhpe_amd import _lib
    assert _lib.lib().d3d_version() >= 100
    assert torch.cuda.is_available()
    maps = open("/proc/self/maps").read()
    assert "libd3d_hip.so" in maps, "native engine not loaded"


@pytest.mark.parametrize("D,depth", [(32, 4), (512, 8)])
def test_time_embedding_table(D, depth):
    """K0: sinusoid -> trunk (Linear, erf-GELU, Linear) -> SiLU -> the 2*depth per-block Linear layers."""
    from diff3dhpe_amd.spec import DenoiserConfig
    from diff3dhpe_amd.engine import Engine
    g = gold(f"temb_D{D}")
    cfg = DenoiserConfig(num_frame=9, embed_dim=D, depth=depth)
    eng = Engine(cfg)
    eng.load_weights(torch_sd(cfg, int(g["seed"])))
    out = eng.time_embedding(torch.from_numpy(g["t"]).cuda())
    assert out.shape == g["per_block"].shape
    e = maxabs(out, g["per_block"])
    print(f"temb D={D}: max-abs {e:.3e}")
    assert e <= 2e-5


DENOISE = [("small_T81", cfg_small(81)), ("full_T27", cfg_full(27)), ("full_T81", cfg_full(81)),
           ("full_T243", cfg_full(243)), ("s2f_T27", cfg_full(27, seq2frame=True)),
           ("notemb_T27", cfg_full(27, with_time_emb=False)), ("small_s2f_T27", cfg_small(27, seq2frame=True))]


PRECS = ["fp32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,cfg", DENOISE, ids=[d[0] for d in DENOISE])
def test_forward_denoise_golden(tag, cfg, prec):
    g = gold("denoise_" + tag)
    B = int(g["B"])
    net, _ = build_product(cfg, int(g["seed"]), precision=prec)
    inp = inputs(B, cfg.num_frame, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1).cuda()
    worst = 0.0
    for t in (999, 443, 0):
        out = net.forward_denoise(xcat, torch.full((B,), t, dtype=torch.long, device="cuda"))
        assert out.shape == g[f"t{t}"].shape
        worst = max(worst, maxabs(out, g[f"t{t}"]))
    out = net.forward_denoise(xcat, torch.from_numpy(g["tmixed_t"]).long().cuda())     # per-row timesteps
    worst = max(worst, maxabs(out, g["tmixed"]))
    print(f"denoise {tag} [{prec}]: max-abs {worst:.3e}")
    assert worst <= GATE


DDIM = [("small_T81_S5", cfg_small(81), True, True), ("full_T81_S9", cfg_full(81), False, True),
        ("full_T243_S9", cfg_full(243), False, True), ("full_T243_S50", cfg_full(243), False, True),
        ("s2f_T27_S9", cfg_full(27, seq2frame=True), True, True),
        ("full_T27_S7_notemb", cfg_full(27, with_time_emb=False), False, True),
        ("small_T81_S5_noclip", cfg_small(81), True, False)]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,cfg,traj,clip", DDIM, ids=[d[0] for d in DDIM])
def test_ddim_loop_golden(tag, cfg, traj, clip, prec):
    g = gold("ddim_" + tag)
    B, S = int(g["B"]), int(g["S"])
    _, diff = build_product(cfg, int(g["seed"]), sampling=S, clip=clip, precision=prec)
    inp = inputs(B, cfg.num_frame, int(g["input_seed"]))
    noise = inp["noise"][:, :1].contiguous() if cfg.seq2frame else inp["noise"]
    clean = torch.zeros_like(noise).cuda()
    x2d = inp["x2d"].cuda()
    if traj:
        loss, y0, rev, x0s = diff(clean, x2d, None, True, False, init_noise=noise.cuda())   # positional, as RUN:196 does
        assert loss is None
        assert rev.shape == g["x_reverse_diffusion"].shape and x0s.shape == g["x_start_est"].shape
        e = max(maxabs(y0, g["y0"]), maxabs(rev, g["x_reverse_diffusion"]), maxabs(x0s, g["x_start_est"]))
        assert torch.equal(y0, x0s[..., -1])       # last step returns the clamped x_start (DIFF:283-285)
    else:
        loss, y0 = diff(clean_3d_pose=clean, noisy_2d_pose=x2d, output_loss=False, init_noise=noise.cuda())
        assert loss is None
        e = maxabs(y0, g["y0"])
    print(f"ddim {tag} [{prec}]: max-abs {e:.3e}")
    assert y0.shape == g["y0"].shape and e <= GATE
    if clip:
        assert y0.abs().max().item() <= 1.0


@pytest.mark.parametrize("prec", PRECS)
def test_repeat_n_and_stochastic_eta(prec):
    g = gold("ddim_small_T27_S4_eta05_rep3")
    cfg = cfg_small(27)
    B, S, R = int(g["B"]), int(g["S"]), int(g["R"])
    _, diff = build_product(cfg, int(g["seed"]), sampling=S, eta=0.5, precision=prec)
    inp = inputs(B * R, 27, int(g["input_seed"]))
    step_noise = torch.stack([hashed(f"eta_noise/{i}", tuple(inp["noise"].shape), 6) for i in range(S)])
    _, y0 = diff(clean_3d_pose=torch.zeros(B, 27, 17, 3).cuda(), noisy_2d_pose=inp["x2d"][:B].cuda(), output_loss=False,
                 repeat_n=R, init_noise=inp["noise"].cuda(), step_noise=step_noise.cuda())
    assert maxabs(y0, g["y0"]) <= GATE


@pytest.mark.parametrize("prec", PRECS)
def test_p_losses_and_q_sample(prec):
    g = gold("plosses_small_T27")
    cfg = cfg_small(27)
    B = int(g["B"])
    _, diff = build_product(cfg, int(g["seed"]), sampling=5, precision=prec)
    inp = inputs(B, 27, int(g["input_seed"]))
    gt = (inp["gt3d"] * float(g["gt_scale"])).cuda()
    t = torch.from_numpy(g["t"]).long().cuda()
    xq = diff.q_sample(gt, t, inp["noise"].cuda())
    assert maxabs(xq, g["q_sample"]) <= 1e-6
    loss = diff.p_losses(gt, inp["x2d"].cuda(), noise=inp["noise"].cuda(), t=t)
    assert maxabs(loss, g["loss"]) <= GATE
    xs, tl = diff.get_noisy_pose(gt, 4)
    assert xs.shape == (B, 27, 17, 3, 4) and tl == [0, 250, 500, 750]


def test_tta_merge_and_mpjpe_kernel():
    from diff3dhpe_amd.engine import tta_mpjpe
    g = gold("evalmath")
    err, cnt, merged = tta_mpjpe(torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["pred_flip"]).cuda(),
                                 torch.from_numpy(g["gt"]).cuda(), torch.from_numpy(g["target_mask"]).cuda(), float(g["scale"]),
                                 g["joints_left"].tolist(), g["joints_right"].tolist(), want_merged=True)
    mask = g["target_mask"].reshape(-1)
    assert np.array_equal(merged.cpu().numpy().reshape(-1, 17, 3)[mask], g["merged"][:, 0])   # bit-exact merge
    assert cnt == int(mask.sum()) * 17
    assert abs(err / cnt - float(g["mpjpe"])) < 1e-6


def test_ddim_index_schedule_is_the_engines_schedule():
    """The schedule the engine steps through is the bit-exact host routine checked (for every S) in the CPU suite;
    here: a 1-step and a 1000-step schedule run end to end without index faults."""
    cfg = cfg_small(9)
    for S in (1, 1000):
        _, diff = build_product(cfg, 1, sampling=S)
        inp = inputs(1, 9, 3)
        _, y0 = diff(clean_3d_pose=torch.zeros(1, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                     init_noise=inp["noise"].cuda())
        assert torch.isfinite(y0).all() and y0.abs().max() <= 1.0
        assert diff.ddim_times()[0] == 999 and diff.ddim_times()[-1] == -1 and len(diff.ddim_times()) == S + 1


@pytest.mark.parametrize("prec", PRECS)
def test_batch_independence_sharding_and_determinism(prec):
    """Domain properties at a size the oracle would take minutes for (T=243, D=512): a sequence's result does not
    depend on its batch neighbours (=> sharding a batch over ranks reproduces the 1-GPU result bit for bit), repeated
    runs are bit-identical, and clamped outputs stay in [-1, 1]."""
    from diff3dhpe_amd import parallel
    cfg = cfg_full(243)
    _, diff = build_product(cfg, 8, sampling=3, precision=prec)
    inp = inputs(6, 243, 77)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    z = torch.zeros_like(nz)
    _, full = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
    _, again = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
    assert torch.equal(full, again)
    parts = []
    for r in range(4):                      # ragged shards 2,2,1,1
        lo, hi = parallel.shard_bounds(6, r, 4)
        _, p = diff(clean_3d_pose=z[lo:hi], noisy_2d_pose=x2d[lo:hi], output_loss=False, init_noise=nz[lo:hi])
        parts.append(p)
    assert torch.equal(torch.cat(parts), full)
    assert full.abs().max().item() <= 1.0 and torch.isfinite(full).all()
    # fewer windows than ranks: an empty shard comes back empty from the sampling loop (the reference's forward() cannot take
    # one -- its view(repeat_n, 0, f, p, -1) is ambiguous --, so evaluate() does not call it for such a rank)
    parts = []
    for r in range(8):
        lo, hi = parallel.shard_bounds(6, r, 8)
        p = diff.ddim_sample_loop(x2d[lo:hi], [hi - lo, 243, 17, 3], init_noise=nz[lo:hi])
        assert tuple(p.shape) == (hi - lo, 243, 17, 3)
        parts.append(p)
    assert parts[-1].shape[0] == 0 and torch.equal(torch.cat(parts), full)


@pytest.mark.parametrize("prec", PRECS)
def test_result_does_not_depend_on_workspace_contents(prec):
    """The caller-owned workspace arrives uninitialised (the GEMM stages the padding rows of its edge tiles, which no
    kernel ever writes): NaN / Inf / random bit patterns in it must not change a single output bit."""
    cfg = cfg_full(27)
    _, diff = build_product(cfg, 8, sampling=3, precision=prec)
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = diff._engine(dev)
    inp = inputs(3, 27, 42)                  # M = 1377 rows: ragged against every tile height
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    ws = eng._workspace(3)
    ws.zero_()
    ref = eng.ddim_sample(x2d, nz).clone()
    for fill in (lambda: ws.fill_(0xFF), lambda: ws.fill_(0x7C),
                 lambda: ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device=dev))):
        fill()
        out = eng.ddim_sample(x2d, nz)
        assert torch.equal(out, ref)
    assert torch.isfinite(ref).all()


def test_large_batch_kernels_match_the_small_batch_path():
    """At B = 32, T = 243 the F16X3 engine runs its large-problem kernels for every GEMM (256x256 launch for the whole
    rounds + 64x256 launch for the remainder rows) and the persistent DMA-staged attention; the golden-vector tests run at
    B <= 6 and never reach them.  Every element is defined to be tile-shape independent, so the large batch must
    reproduce the small-batch results bit for bit -- and twice in a row."""
    cfg = cfg_full(243)
    _, diff = build_product(cfg, 8, sampling=1, precision="f16x3")
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = diff._engine(dev)
    inp = inputs(32, 243, 5)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    big = eng.ddim_sample(x2d, nz).clone()
    again = eng.ddim_sample(x2d, nz)
    assert torch.equal(big, again)
    for lo in (0, 13, 26):                   # ragged chunks of the small-problem path
        hi = min(lo + 13, 32)
        part = eng.ddim_sample(x2d[lo:hi].contiguous(), nz[lo:hi].contiguous())
        assert torch.equal(part, big[lo:hi]), f"samples {lo}:{hi} differ by {(part - big[lo:hi]).abs().max().item():.3e}"
    assert torch.isfinite(big).all() and big.abs().max().item() <= 1.0
    # one stream: the whole batch in one carve-up, and the persistent GEMM walks cut their partly filled last round into row slices
    # (the two-stream default keeps those tiles whole): same bits
    eng.set_option("streams", 1)
    assert torch.equal(eng.ddim_sample(x2d, nz), big)
    eng.set_option("streams", 2)
    # T = 81: three key tiles per unit -- the other instantiation of the persistent attention kernel (needs >= 1024 units)
    cfg = cfg_full(81)
    _, diff = build_product(cfg, 8, sampling=2, precision="f16x3")
    eng = diff._engine(dev)
    inp = inputs(12, 81, 6)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    big = eng.ddim_sample(x2d, nz).clone()
    assert torch.equal(big, eng.ddim_sample(x2d, nz))
    for lo in (0, 5, 10):
        hi = min(lo + 5, 12)
        assert torch.equal(eng.ddim_sample(x2d[lo:hi].contiguous(), nz[lo:hi].contiguous()), big[lo:hi])
    # T = 27 (BASELINE configs[4]): temporal groups of 27 tokens run the wave-private persistent kernel with six units per workgroup
    # and four patch passes (>= 4096 units), spatial groups the eight-unit form; chunks of 13 run the one-unit-per-workgroup kernel
    cfg = cfg_full(27)
    _, diff = build_product(cfg, 8, sampling=2, precision="f16x3")
    eng = diff._engine(dev)
    inp = inputs(40, 27, 7)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    big = eng.ddim_sample(x2d, nz).clone()
    assert torch.equal(big, eng.ddim_sample(x2d, nz))
    for lo in (0, 13, 26, 39):
        hi = min(lo + 13, 40)
        assert torch.equal(eng.ddim_sample(x2d[lo:hi].contiguous(), nz[lo:hi].contiguous()), big[lo:hi])


@pytest.mark.parametrize("prec", PRECS)
def test_engine_matches_oracle_on_fresh_seeded_inputs(prec):
    """HIP path vs the CPU oracle on inputs no fixture holds (small enough for the oracle to finish in seconds)."""
    from oracle import d3d_oracle as orc
    for cfg, B, S in ((cfg_small(27), 5, 6), (cfg_full(27), 3, 4), (cfg_full(9, seq2frame=True), 2, 3)):
        _, diff = build_product(cfg, 31, sampling=S, precision=prec)
        inp = inputs(B, cfg.num_frame, 555)
        noise = inp["noise"][:, :1].contiguous() if cfg.seq2frame else inp["noise"]
        _, y0 = diff(clean_3d_pose=torch.zeros_like(noise).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                     init_noise=noise.cuda())
        ref = orc.ddim_sample_loop(torch_sd(cfg, 31), orc.diffusion_tables("cosine", 1000), inp["x2d"], noise,
                                   num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth, seq2frame=cfg.seq2frame)
        assert maxabs(y0, ref) <= GATE


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_engine_matches_oracle_on_other_widths(prec):
    """Widths and ratios the golden fixtures do not hold.  In F16X3 mode D=256 / 4 heads runs the plane-resident, LayerNorm-folded
    flow WITHOUT the post-norm GEMM form (that exists for D=512 only: fc2 -> fp32 -> row kernel); D=512 with mlp_ratio 4
    runs the post-norm form over K=2048; 2 heads of 64 is the narrowest width the fp16-MFMA attention takes."""
    from oracle import d3d_oracle as orc
    from diff3dhpe_amd.spec import DenoiserConfig
    for cfg, B, S in ((DenoiserConfig(num_frame=27, embed_dim=256, depth=4, num_heads=4), 3, 3),
                      (DenoiserConfig(num_frame=9, embed_dim=512, depth=2, mlp_ratio=4.0), 2, 2),
                      (DenoiserConfig(num_frame=27, embed_dim=128, depth=2, num_heads=2), 2, 3)):
        _, diff = build_product(cfg, 47, sampling=S, precision=prec)
        inp = inputs(B, cfg.num_frame, 556)
        _, y0 = diff(clean_3d_pose=torch.zeros_like(inp["noise"]).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                     init_noise=inp["noise"].cuda())
        ref = orc.ddim_sample_loop(torch_sd(cfg, 47), orc.diffusion_tables("cosine", 1000), inp["x2d"], inp["noise"],
                                   num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth, heads=cfg.num_heads)
        assert maxabs(y0, ref) <= GATE, (cfg.embed_dim, cfg.num_heads, cfg.mlp_ratio)


@pytest.mark.parametrize("off", [2.0, 8.0])
def test_folded_layernorm_statistics_with_offset_rows(off):
    """The LayerNorm folded into the qkv / fc1 GEMMs takes its row variance as sum(x^2)/D - mean^2 from the producer's (sum,
    sum of squares) partials -- a one-pass form whose relative error grows like eps * (1 + mean^2/var).  Rows of the stream are
    LayerNorm outputs (|mean| well below the standard deviation for any sane checkpoint); this test pushes the post-norm biases
    and the position embeddings so that |mean| = 2 and 8 standard deviations and checks the gate against the oracle (two-pass
    statistics) still holds with a wide margin.  DESIGN.md section 2 states the bound; the guard bit that covers the rest of
    the range is tested in tests/test_gpu_round3.py::test_range_guard_row_statistics_bit."""
    from oracle import d3d_oracle as orc
    cfg = cfg_full(9)
    sd = torch_sd(cfg, 77)
    for k in ("Spatial_norm.bias", "Temporal_norm.bias"):
        sd[k] = sd[k] + off * sd[k.replace("bias", "weight")].abs().mean()
    import diff3dhpe_amd as d3d
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=9, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8, mlp_ratio=2.0,
                                      qkv_bias=True, qk_scale=None, drop_path_rate=0.1)
    net.load_state_dict(sd, strict=True)
    net.precision = "f16x3"
    net = net.cuda()
    inp = inputs(3, 9, 557)
    xcat = torch.cat([inp["x2d"], inp["noise"]], dim=-1)
    t = torch.tensor([999, 400, 3])
    out = net.forward_denoise(xcat.cuda(), t.cuda())
    ref = orc.forward_denoise(sd, xcat, t, depth=8)
    net.precision = "fp32"                         # the plain dataflow: two-pass row kernels, no folded statistics
    e, e32 = maxabs(out, ref), maxabs(net.forward_denoise(xcat.cuda(), t.cuda()), ref)
    print(f"row mean = {off} std: folded F16X3 flow vs oracle {e:.3e}; FP32 flow vs oracle {e32:.3e}")
    assert e <= GATE / 4


@pytest.mark.parametrize("prec", PRECS)
def test_evaluate_harness_flip_tta(prec):
    from diff3dhpe_amd.evaluate import evaluate, flip_2d, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    from oracle import d3d_oracle as orc
    cfg = cfg_small(27)
    _, diff = build_product(cfg, 12, sampling=3, precision=prec)
    inp = inputs(4, 27, 99)
    mask = torch.ones(4, 27, dtype=torch.bool)
    mask[3, 20:] = False
    nz_f = hashed("flipnoise", tuple(inp["noise"].shape), 1)
    res = evaluate(diff, [{"inputs_2d": inp["x2d"], "inputs_3d": inp["gt3d"], "target_mask": mask,
                           "init_noise": inp["noise"], "init_noise_flip": nz_f}], scale=1.3, verbose=False)
    sd, tabs = torch_sd(cfg, 12), orc.diffusion_tables("cosine", 1000)
    kw = dict(num_timesteps=1000, sampling_timesteps=3, depth=cfg.depth)
    p = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], **kw)
    pf = orc.ddim_sample_loop(sd, tabs, flip_2d(inp["x2d"], H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT), nz_f, **kw)
    merged = orc.merge_flip_tta(p, pf, 1.3, mask)
    gtm = inp["gt3d"].view(-1, 17, 3)[mask.view(-1)].unsqueeze(1)
    assert abs(res["mpjpe_mm"] - orc.mpjpe(merged, gtm).item() * 1000) < 0.05
    assert res["frames"] == int(mask.sum())


def test_hipgraph_replay_matches_eager_launches():
    """cfg4 mechanics: the captured S-step graph must reproduce the eager launch sequence bit for bit, across replays,
    batch sizes and a weight reload (which must drop the stale graphs)."""
    cfg = cfg_full(27)
    net, diff = build_product(cfg, 21, sampling=5, precision="f16x3")
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    z = lambda n: torch.zeros(n, 27, 17, 3, device="cuda")
    outs = {}
    for B in (3, 1):
        inp = inputs(B, 27, 60 + B)
        x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
        eng.set_graph_mode(False)
        _, eager = diff(clean_3d_pose=z(B), noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
        eng.set_graph_mode(True)
        for rep in range(3):
            _, g = diff(clean_3d_pose=z(B), noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            assert torch.equal(g, eager), (B, rep)
        outs[B] = (x2d, nz, eager)
    # new weights -> engines re-commit -> graphs dropped and re-captured with the new tensors
    net.load_state_dict(torch_sd(cfg, 22))
    eng2 = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng2.set_graph_mode(True)
    x2d, nz, old = outs[3]
    _, g_new = diff(clean_3d_pose=z(3), noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
    eng2.set_graph_mode(False)
    _, e_new = diff(clean_3d_pose=z(3), noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
    assert torch.equal(g_new, e_new) and not torch.equal(g_new, old)


def test_sequence_windows_on_device_bit_exact():
    """d3d_window_gather (index work) must be bit-exact with the oracle's window table for ragged / short / exact lengths."""
    from diff3dhpe_amd.engine import window_gather
    from oracle import d3d_oracle as orc
    kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    g = gold("chunks")
    for n, T in [(700, 243), (243, 243), (486, 243), (487, 243), (100, 27), (81, 27), (20, 27), (1, 9)]:
        rng = np.random.RandomState(n * 1000 + T)
        p2 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32))
        p3 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32))
        w, m = window_gather(p2.cuda(), T)
        wf = window_gather(p2.cuda(), T, True, kl, kr, want_mask=False)
        w3 = window_gather(p3.cuda(), T, want_mask=False)
        ow, om = orc.gather_windows(p2, T)
        owf, _ = orc.gather_windows(p2, T, True, kl, kr)
        ow3, _ = orc.gather_windows(p3, T)
        assert torch.equal(w.cpu(), ow) and torch.equal(wf.cpu(), owf) and torch.equal(w3.cpu(), ow3), (n, T)
        assert torch.equal(m.cpu(), om) and np.array_equal(m.cpu().numpy(), g[f"n{n}_T{T}/mask"]), (n, T)


@pytest.mark.parametrize("prec", PRECS)
def test_evaluate_sequence_end_to_end(prec):
    """A 70-frame video through windows -> DDIM (flip TTA) -> merge -> masked MPJPE, against the oracle doing the same."""
    from diff3dhpe_amd.evaluate import evaluate_sequence, H36M_JOINTS_LEFT as JL, H36M_JOINTS_RIGHT as JR
    from oracle import d3d_oracle as orc
    cfg = cfg_small(27)
    _, diff = build_product(cfg, 14, sampling=3, precision=prec)
    n = 70
    rng = np.random.RandomState(5)
    p2 = torch.from_numpy(np.clip(rng.normal(0, 0.4, (n, 17, 2)), -1, 1).astype(np.float32))
    p3 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32))
    nz = hashed("seqnoise", (3, 27, 17, 3), 2)
    nzf = hashed("seqnoise_f", (3, 27, 17, 3), 3)
    res = evaluate_sequence(diff, p2, p3, num_frames=27, scale=2.0, init_noise=nz, init_noise_flip=nzf)
    sd, tabs = torch_sd(cfg, 14), orc.diffusion_tables("cosine", 1000)
    kw = dict(num_timesteps=1000, sampling_timesteps=3, depth=cfg.depth)
    w, m = orc.gather_windows(p2, 27)
    wf, _ = orc.gather_windows(p2, 27, True, JL, JR)
    g3, _ = orc.gather_windows(p3, 27)
    merged = orc.merge_flip_tta(orc.ddim_sample_loop(sd, tabs, w, nz, **kw), orc.ddim_sample_loop(sd, tabs, wf, nzf, **kw), 2.0, m)
    gtm = g3.reshape(-1, 17, 3)[m.reshape(-1)].unsqueeze(1)
    assert res["frames"] == n and abs(res["mpjpe_mm"] - orc.mpjpe(merged, gtm).item() * 1000) < 0.05


def test_c_abi_error_behaviour_on_device():
    """Error codes of the compute entry points (no exception crosses the ABI; nothing silently falls back)."""
    import ctypes as C
    from diff3dhpe_amd import _lib
    from diff3dhpe_amd.engine import Engine
    L = _lib.lib()
    cfg = cfg_small(9)
    eng = Engine(cfg, precision="f16x3")
    x2d = torch.zeros(2, 9, 17, 2, device="cuda")
    y = torch.zeros(2, 9, 17, 3, device="cuda")
    out = torch.empty_like(y)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    # weights not committed yet
    assert L.d3d_ddim_sample(eng._h, p(x2d), p(y), None, p(out), None, None, 2, p(ws), ws.numel(), None) == -2
    eng.load_weights(torch_sd(cfg, 1))
    # schedule not set
    need = L.d3d_workspace_bytes(eng._h, 2)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert L.d3d_ddim_sample(eng._h, p(x2d), p(y), None, p(out), None, None, 2, p(ws), ws.numel(), None) == -2
    assert b"schedule" in L.d3d_last_error()
    tabs = build_product(cfg, 1, sampling=3)[1]
    eng.set_schedule(tabs.alphas_cumprod, tabs.sqrt_one_minus_alphas_cumprod, 3, 0.5, True)
    # eta != 0 without step noise, too-small workspace, null tensors, bad batch
    assert L.d3d_ddim_sample(eng._h, p(x2d), p(y), None, p(out), None, None, 2, p(ws), ws.numel(), None) == -1
    assert L.d3d_ddim_sample(eng._h, p(x2d), p(y), p(y), p(out), None, None, 2, p(ws), need // 2, None) == -4
    assert L.d3d_ddim_sample(eng._h, None, p(y), p(y), p(out), None, None, 2, p(ws), ws.numel(), None) == -1
    assert L.d3d_ddim_sample(eng._h, p(x2d), p(y), p(y), p(out), None, None, 0, p(ws), ws.numel(), None) == -1
    assert L.d3d_denoise(eng._h, p(x2d), p(y), 5, None, 0, p(out), 2, p(ws), ws.numel(), None) == -1      # y_frames must be 1 or T
    assert L.d3d_denoise(eng._h, p(x2d), p(y), 9, None, 0, p(out), 2, p(ws), ws.numel(), None) == -1      # times required
    with pytest.raises(_lib.D3DError):
        Engine(cfg, precision="f16x3").ddim_sample(x2d, y)       # python wrapper: set_schedule() first
    # a healthy call still works afterwards
    eng.set_schedule(tabs.alphas_cumprod, tabs.sqrt_one_minus_alphas_cumprod, 3, 0.0, True)
    res = eng.ddim_sample(x2d, y)
    assert torch.isfinite(res).all()


@pytest.mark.parametrize("prec", PRECS)
def test_dataset_adaptor_feeds_evaluate(prec):
    """Section 8f row 4 end to end: windows from diff3dhpe_amd.data (bit-equal to the reference loader, tests/test_data_adaptor.py)
    through evaluate() -- two samplings per window, merge, de-normalise by the data set's scale, masked MPJPE (RUN:562-606) --
    against the oracle doing the same on the same windows."""
    from diff3dhpe_amd.data import EvalData, MocapMeta
    from diff3dhpe_amd.evaluate import evaluate
    from diff3dhpe_amd.synth import synth_mocap, SYNTH_JOINTS_LEFT as JL, SYNTH_JOINTS_RIGHT as JR
    from oracle import d3d_oracle as orc
    pos, cams, kp, meta = synth_mocap(0)
    ed = EvalData(MocapMeta(pos, cams, JL, JR), kp, meta["keypoints_symmetry"], ["S9"], 27)      # 8 windows
    cfg = cfg_small(27)
    _, diff = build_product(cfg, 31, sampling=3, precision=prec)
    batches = []
    for i, b in enumerate(ed.batches(5)):
        n = b["inputs_2d"].shape[0]
        b["init_noise"] = hashed(f"dsn{i}", (n, 27, 17, 3), 4)
        b["init_noise_flip"] = hashed(f"dsf{i}", (n, 27, 17, 3), 5)
        batches.append(b)
    res = evaluate(diff, batches, scale=ed.scale, joints_left=ed.joints_left, joints_right=ed.joints_right, verbose=False)
    sd, tabs = torch_sd(cfg, 31), orc.diffusion_tables("cosine", 1000)
    kw = dict(num_timesteps=1000, sampling_timesteps=3, depth=cfg.depth)
    err, cnt = 0.0, 0
    for b in batches:
        p = orc.ddim_sample_loop(sd, tabs, b["inputs_2d"], b["init_noise"], **kw)
        pf = orc.ddim_sample_loop(sd, tabs, b["inputs_2d_flip"], b["init_noise_flip"], **kw)
        merged = orc.merge_flip_tta(p, pf, ed.scale, b["target_mask"])
        gtm = b["inputs_3d"].view(-1, 17, 3)[b["target_mask"].view(-1)].unsqueeze(1)
        err += orc.mpjpe(merged, gtm).item() * gtm.shape[0]
        cnt += gtm.shape[0]
    assert res["frames"] == cnt == int(sum(int(b["target_mask"].sum()) for b in batches))
    assert abs(res["mpjpe_mm"] - err / cnt * 1000) < 0.05


@pytest.mark.parametrize("prec", PRECS)
def test_rng_is_consumed_as_the_reference_consumes_it(prec):
    """Without supplied noise a sampling draws randn(target_shape) once and then S - 1 more tensors of that shape from the global
    generator, eta or not (DIFF:275, 293): the next sampling of an unchanged runner starts from the same generator state."""
    cfg = cfg_small(27)
    _, diff = build_product(cfg, 12, sampling=4, precision=prec)
    x2d = inputs(2, 27, 7)["x2d"].cuda()
    shape = (2, 27, 17, 3)
    torch.manual_seed(1234)
    first = torch.randn(shape, device="cuda")
    for _ in range(3):
        torch.randn(shape, device="cuda")
    expect_next = torch.randn(shape, device="cuda")
    torch.manual_seed(1234)
    _, y_auto = diff(clean_3d_pose=torch.zeros(shape, device="cuda"), noisy_2d_pose=x2d, output_loss=False)
    got_next = torch.randn(shape, device="cuda")
    assert torch.equal(got_next, expect_next)
    _, y_given = diff(clean_3d_pose=torch.zeros(shape, device="cuda"), noisy_2d_pose=x2d, output_loss=False, init_noise=first)
    assert torch.equal(y_auto, y_given)


def test_f16x3_range_guard():
    """Operands beyond the fp16 range of the F16X3 planes (|x| > 8188, a non-finite weight) must raise the sticky flags -- and only
    they; a large finite weight only lowers its matrix's plane scale (parity at such weights: the trained-like goldens)."""
    from diff3dhpe_amd import _lib
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=1)
    inp = inputs(2, 27, 3)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    dev = torch.device("cuda", torch.cuda.current_device())

    def run(mutate):
        sd = torch_sd(cfg, 8)
        mutate(sd)
        net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=27, embed_dim=512, depth=1)
        net.load_state_dict(sd)
        net.precision = "f16x3"
        net.range_check = False          # this test reads the engine's raw word itself (the default reading: tests/test_gpu_round5.py)
        diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=2, clip_denoised=True).eval().to(dev)
        eng = diff._engine(dev)
        eng.range_flags(clear=True)
        diff(clean_3d_pose=torch.zeros_like(nz), noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
        return eng
    eng = run(lambda sd: None)
    assert eng.range_flags() == 0
    eng.check_range()                                                            # healthy model: no exception

    def big_w(sd):
        sd["STEblocks.0.mlp.fc1.weight"][3, 5] = 20.0                            # beyond 15.99: the matrix gets a smaller plane scale
    eng = run(big_w)
    assert eng.range_flags() == 0                                                # (2^11 instead of 2^12: nothing is clamped)

    def inf_w(sd):
        sd["STEblocks.0.mlp.fc1.weight"][3, 5] = float("inf")                    # not representable at any scale
    eng = run(inf_w)
    assert eng.range_flags(clear=False) & _lib.RANGE_WEIGHT
    with pytest.raises(_lib.D3DError, match="range"):
        eng.check_range()

    def big_x(sd):
        sd["fusion_layer.bias"] += 1.0e4                                         # residual stream ~1e4: |8 x| > 65504
    eng = run(big_x)
    f = eng.range_flags()
    assert f & _lib.RANGE_ACT and not (f & _lib.RANGE_WEIGHT)
    assert eng.range_flags() == 0                                                # cleared by the read


def test_hipgraph_replay_long_chain_T243():
    """BASELINE configs[3] at its own size: T=243, 50 DDIM steps, B=4 -- one captured graph of 50 x 82 launches replayed twice,
    bit-identical to the eager launch sequence."""
    cfg = cfg_full(243)
    _, diff = build_product(cfg, 5, sampling=50, precision="f16x3")
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    inp = inputs(4, 243, 77)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    z = torch.zeros(4, 243, 17, 3, device="cuda")
    eng.set_graph_mode(False)
    _, eager = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
    eng.set_graph_mode(True)
    try:
        for rep in range(2):
            _, g = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            assert torch.equal(g, eager), rep
    finally:
        eng.set_graph_mode(False)
    assert torch.isfinite(eager).all() and float(eager.abs().max()) <= 1.0


def test_bench_two_ranks_on_one_device():
    """The N>1 code path of bench.py (sharding, all-gather of predictions, max-over-ranks timing, rank-0 JSON) run as two
    torch.distributed ranks that share cuda:0 through gloo (RCCL cannot form a communicator on one device).  The gathered
    MPJPE must equal the 1-rank value for the same global batch (per-sample noise is a slice of the global tensor).
    Runs the DEFAULT kernels.  Two processes on one GPU once showed run-to-run different single outputs of the head kernel; the
    cause was narrowed to the instruction stream of its 3-row dot product (the mechanism below it is not identified:
    experiments/NOTES.md), the kept stream is pinned by tests/test_abi_host.py::test_head_kernel_instruction_stream_..., and
    experiments/two_rank_repeat.sh is the many-iteration form of this test (one step here barely sees a 1-in-60 event)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, D3D_BENCH_ONE_DEVICE="1", D3D_DIST_BACKEND="gloo")
    common = ["--steps", "1", "--warmup", "0", "--frames", "27", "--sampling", "3", "--no-cpu-baseline", "--no-extras"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "3"] + common,
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    env1 = dict(os.environ)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "6"] + common,
                         capture_output=True, text=True, cwd=ROOT, timeout=600, env=env1)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["config"]["global_batch"] == 6 and line1["config"]["global_batch"] == 6
    assert line2["mpjpe_vs_synthetic_gt"] == line1["mpjpe_vs_synthetic_gt"]
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "headline_under", "ranks"):
        assert key in line2
    rk = line2["ranks"]          # per-rank spread and the event-timed exchange step (what the first real SCALE record will carry)
    assert rk["ms_per_step_min_over_ranks"] <= rk["ms_per_step_max_over_ranks"] == line2["ms_per_step"]
    assert rk["allgather_ms_per_step_max_over_ranks"] > 0 and rk["allgather_bytes_per_rank"] == 3 * 27 * 17 * 3 * 4
    assert line1["headline_under"] == "2-stream" and "ranks" not in line1 and "timed_in" in line1["roofline"]


@pytest.mark.skipif(not __import__("os").environ.get("D3D_SLOW_TESTS"),
                    reason="~3 min of GPU: set D3D_SLOW_TESTS=1 -- the many-iteration form of test_bench_two_ranks_on_one_device")
def test_two_ranks_on_one_device_many_iterations():
    """Opt-in.  The deviation a second process on the GPU once caused in the head kernel showed in about 1 launch of 60: one
    two-rank run (3 samplings x 3 steps) barely sees such an event, twenty of them at T = 243 (9 steps each) would have shown it
    with probability > 0.99.  Every run must report the same MPJPE, bit for bit (experiments/two_rank_repeat.sh as a test)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, D3D_BENCH_ONE_DEVICE="1", D3D_DIST_BACKEND="gloo")
    seen = set()
    for i in range(20):
        run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(29700 + i), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "2",
                              "--warmup", "0", "--frames", "243", "--sampling", "9", "--no-cpu-baseline", "--no-extras", "--no-selfcheck",
                              "--profile-steps", "0"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
        assert run.returncode == 0, run.stderr[-2000:]
        seen.add(json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])["mpjpe_vs_synthetic_gt"])
    assert len(seen) == 1, seen


def test_bench_rccl_collectives_on_a_one_rank_group():
    """bench.py's N > 1 communication path on RCCL itself, on a one-GPU box: `torchrun --nproc-per-node 1 bench.py --gpus 1`
    with D3D_FORCE_DIST=1 executes init_process_group("nccl", device_id=...), barrier, all_gather_into_tensor and
    all_reduce(MAX) on a one-rank communicator (RUN:216-218 replacement; the launcher starts before any GPU call of the child)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, D3D_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("D3D_DIST_BACKEND", None)
    common = ["--steps", "2", "--warmup", "1", "--frames", "27", "--sampling", "3", "--no-cpu-baseline", "--no-extras"]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "4"] + common,
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1 and line["dist"]["forced_single_rank_group"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "4"] + common,
                           capture_output=True, text=True, cwd=ROOT, timeout=600, env={k: v for k, v in os.environ.items() if k != "D3D_FORCE_DIST"})
    assert plain.returncode == 0, plain.stderr[-2000:]
    ref = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert "dist" not in ref
    assert line["mpjpe_vs_synthetic_gt"] == ref["mpjpe_vs_synthetic_gt"]        # the gathered prediction is the local one


def test_bench_graph_flag_times_the_main_leg_under_replay():
    """`bench.py --graph` (BASELINE configs[3] protocol): the main leg runs as hipGraph replays, prints a complete line whose
    prediction error equals the eager run's (same values), and says that per-kernel timing was off."""
    import json, os, subprocess, sys
    from conftest import ROOT
    common = ["--gpus", "1", "--batch", "4", "--steps", "2", "--warmup", "1", "--frames", "27", "--sampling", "5",
              "--no-cpu-baseline", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k != "D3D_FORCE_DIST"}
    lines = []
    for extra in ([], ["--graph"]):
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, capture_output=True, text=True,
                             cwd=ROOT, timeout=600, env=env)
        assert run.returncode == 0, run.stderr[-2000:]
        lines.append(json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1]))
    eager, graph = lines
    assert graph["graph_replay"] and "graph_replay" not in eager
    assert graph["headline_under"] == "2-stream+graph" and eager["headline_under"] == "2-stream"
    assert graph["mpjpe_vs_synthetic_gt"] == eager["mpjpe_vs_synthetic_gt"]
    assert graph["selfcheck_batch_vs_pair_bit_identical"] and graph["roofline"]["frac"] > 0
    assert graph["roofline"]["kernel"] == eager["roofline"]["kernel"]     # the roofline comes from the separate profiled pass in both


def test_c_abi_allgather_on_a_one_rank_rccl_communicator():
    """d3d_allgather_pred over a communicator made with RCCL's own C API (ncclGetUniqueId / ncclCommInitRank, world size 1) from
    the RCCL library torch has loaded: the entry resolves ncclAllGather at run time and the gathered buffer equals the send
    buffer.  Runs in a child process (a communicator beside torch's own state is nothing to leave behind in the test process)."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import ctypes as C, glob, os, sys, torch
        sys.path.insert(0, %r)
        from diff3dhpe_amd import _lib
        torch.cuda.init(); torch.cuda.set_device(0)
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so"]
        R = C.CDLL(cands[0], mode=C.RTLD_GLOBAL)
        class UID(C.Structure):
            _fields_ = [("b", C.c_char * 128)]
        uid = UID()
        assert R.ncclGetUniqueId(C.byref(uid)) == 0
        comm = C.c_void_p()
        R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
        assert R.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        x = torch.randn(4, 27, 17, 3, device="cuda")
        y = torch.zeros_like(x)
        st = torch.cuda.current_stream().cuda_stream
        rc = _lib.lib().d3d_allgather_pred(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), C.c_void_p(st))
        torch.cuda.synchronize()
        assert rc == 0, _lib.lib().d3d_last_error()
        assert torch.equal(x, y)
        R.ncclCommDestroy.argtypes = [C.c_void_p]
        R.ncclCommDestroy(comm)
        print("allgather ok")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert run.returncode == 0 and "allgather ok" in run.stdout, run.stderr[-2000:]


def test_per_kernel_trace_is_reproducible_and_does_not_change_results():
    """d3d_engine_set_trace: a checksum entry per buffer the block flow writes, in launch order.  Two samplings of the same inputs
    give the same (tag, checksum) list, the list has an entry for every kernel kind of every block of every forward, the XCD
    views of a buffer agree with each other, and the sampled prediction is bitwise the one of an untraced run."""
    cfg = cfg_full(27)
    S, B = 3, 2
    _, diff = build_product(cfg, 5, sampling=S, precision="f16x3")
    inp = inputs(B, 27, 77)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    plain = eng.ddim_sample(x2d, nz).clone()
    eng.set_trace(4096, 1)
    a = eng.ddim_sample(x2d, nz)
    ta = eng.trace_read()
    b = eng.ddim_sample(x2d, nz)
    tb = eng.trace_read()
    assert torch.equal(a, plain) and torch.equal(b, plain)
    assert ta == tb and len(ta) >= S * (5 * 2 * cfg.depth + 3)
    kinds = {(t >> 4) & 0xF for t, _ in ta}
    assert {3, 4, 5, 6, 8} <= kinds                    # attention (fused with its qkv GEMM), proj, fc1, fc2 + post-norm, head
    eng.set_option("fused_temporal", 0)                # the two-kernel flow of the temporal blocks writes (and traces) q / k / v planes
    eng.ddim_sample(x2d, nz)
    assert 2 in {(t >> 4) & 0xF for t, _ in eng.trace_read()}
    eng.set_option("fused_temporal", 1)
    fwds = {(t >> 16) & 0xFFF for t, _ in ta}
    assert len(fwds) == S
    eng.set_trace(32768, 8)                            # eight views per buffer: every view must report the same words
    eng.ddim_sample(x2d, nz)
    tv = eng.trace_read()
    by_tag = {}
    for t, s in tv:
        by_tag.setdefault(t & 0x0FFFFFFF, set()).add(s)
    assert by_tag and all(len(v) == 1 for v in by_tag.values())
    eng.set_trace(0)
    assert torch.equal(eng.ddim_sample(x2d, nz), plain)
