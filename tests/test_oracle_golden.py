"""The CPU oracle (oracle/d3d_oracle.py) against the golden vectors captured from the real reference
(oracle/gen_golden.py).  This is the pin that lets the GPU tests trust the oracle where no fixture exists."""
import numpy as np
import pytest
import torch

from conftest import gold
from helpers import cfg_small, cfg_full, torch_sd, inputs, hashed
from oracle import d3d_oracle as orc

TOL = 2e-6   # oracle vs reference was bit-exact on the generating host; allow libm/BLAS drift on other hosts


def test_schedule_tables():
    g = gold("schedules")
    tabs = orc.diffusion_tables("cosine", 1000)
    for k, v in tabs.items():
        assert np.array_equal(v.numpy(), g["cosine/" + k]), k
    for sched in ("linear", "logcosine"):
        t = orc.diffusion_tables(sched, 1000)
        for k in ("betas", "alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_alphas_cumprod"):
            assert np.array_equal(t[k].numpy(), g[f"{sched}/{k}"]), (sched, k)
    assert np.array_equal(orc.diffusion_tables("cosine", 100)["alphas_cumprod"].numpy(), g["cosine100/alphas_cumprod"])
    # known answers quoted in SURVEY.md section 8(a2)
    ac = tabs["alphas_cumprod"].numpy()
    assert np.isclose(ac[999], 2.428766965e-09, rtol=1e-6) and np.isclose(ac[887], 3.015837632e-02, rtol=1e-6)
    assert np.isclose(ac[110], 9.661540985e-01, rtol=1e-6) and np.isclose(ac[0], 9.999586940e-01, rtol=1e-6)
    with pytest.raises(ValueError):
        orc.beta_schedule("quadratic", 10)


def test_ddim_times_all_S():
    g = gold("ddim_times_N1000")
    flat, offs = g["flat"], g["offsets"]
    for S in range(1, 1001):
        ref = flat[offs[S - 1]:offs[S]].tolist()
        assert orc.ddim_times(1000, S) == ref, S
        assert orc.ddim_times_scalar(1000, S) == ref, S
    assert orc.ddim_times(1000, 5) == [999, 799, 599, 399, 199, -1]
    for key in ("N100_S20", "N100_S100", "N50_S7", "N200_S33"):
        N, S = (int(x[1:]) for x in key.split("_"))
        assert orc.ddim_times_scalar(N, S) == g[key].tolist()


@pytest.mark.parametrize("D,depth", [(32, 4), (512, 8)])
def test_time_embedding(D, depth):
    from diff3dhpe_amd.spec import DenoiserConfig
    g = gold(f"temb_D{D}")
    sd = torch_sd(DenoiserConfig(num_frame=9, embed_dim=D, depth=depth), int(g["seed"]))
    t = torch.from_numpy(g["t"]).long()
    assert np.abs(orc.sinusoid(t, D).numpy() - g["sinusoid"]).max() <= TOL
    trunk = orc.time_trunk(sd, t, D)
    assert np.abs(trunk.numpy() - g["trunk"]).max() <= TOL
    import torch.nn.functional as F
    for k in range(2 * depth):
        p = f"{'TTEblocks' if k & 1 else 'STEblocks'}.{k // 2}.time_mlp.1"
        v = F.linear(F.silu(trunk), sd[p + ".weight"], sd[p + ".bias"])
        assert np.abs(v.numpy() - g["per_block"][:, k]).max() <= TOL, k


def test_attention_and_blocks():
    from diff3dhpe_amd.spec import DenoiserConfig
    g = gold("attention")
    for tag, D, N, G in (("spatial_D512", 512, 17, 2), ("spatial_D32", 32, 17, 4), ("temporal_D512_T27", 512, 27, 1),
                         ("temporal_D512_T81", 512, 81, 1), ("temporal_D512_T243", 512, 243, 1), ("temporal_D32_T81", 32, 81, 2)):
        sd = torch_sd(DenoiserConfig(num_frame=9, embed_dim=D, depth=1), 2)
        x = hashed("attn_in/" + tag, (G, N, D), 2, 1.5)
        pfx = ("STEblocks.0" if tag.startswith("spatial") else "TTEblocks.0") + ".attn"
        assert np.abs(orc.grand_attention(sd, pfx, x, 8).numpy() - g[tag]).max() <= TOL, tag
    g = gold("blocks")
    for tag, D, shape in (("ste_D512", 512, (1, 4, 17, 512)), ("tte_D512_T81", 512, (1, 81, 2, 512)),
                          ("ste_D32", 32, (2, 9, 17, 32)), ("tte_D32_T27", 32, (2, 27, 17, 32))):
        sd = torch_sd(DenoiserConfig(num_frame=shape[1], embed_dim=D, depth=1), 3)
        x = hashed("block_in/" + tag, shape, 3, 1.2)
        temb = hashed("block_temb/" + tag, (shape[0], 2 * D), 3)
        sp = tag.startswith("ste")
        y = orc.mixste_block(sd, "STEblocks.0" if sp else "TTEblocks.0", x, sp, temb, 8)
        assert np.abs(y.numpy() - g[tag + "/block"]).max() <= TOL, tag


DENOISE = [("small_T81", cfg_small(81)), ("full_T27", cfg_full(27)), ("full_T81", cfg_full(81)),
           ("s2f_T27", cfg_full(27, seq2frame=True)), ("notemb_T27", cfg_full(27, with_time_emb=False)),
           ("small_s2f_T27", cfg_small(27, seq2frame=True)), ("full_T243", cfg_full(243)),
           ("s2f_notemb_T27", cfg_full(27, seq2frame=True, with_time_emb=False))]


@pytest.mark.parametrize("tag,cfg", DENOISE, ids=[d[0] for d in DENOISE])
def test_forward_denoise(tag, cfg):
    g = gold("denoise_" + tag)
    B = int(g["B"])
    sd = torch_sd(cfg, int(g["seed"]))
    inp = inputs(B, cfg.num_frame, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1)
    ts = (443, "mixed") if tag == "full_T243" else (999, 443, 0, "mixed")   # keep the CPU suite short
    for t in ts:
        if t == "mixed":
            tv, ref = torch.from_numpy(g["tmixed_t"]).long(), g["tmixed"]
        else:
            tv, ref = torch.full((B,), t, dtype=torch.long), g[f"t{t}"]
        out = orc.forward_denoise(sd, xcat, tv, depth=cfg.depth, seq2frame=cfg.seq2frame)
        assert np.abs(out.numpy() - ref).max() <= TOL, (tag, t)


DDIM = [("small_T81_S5", cfg_small(81), True, True), ("full_T81_S9", cfg_full(81), False, True),
        ("s2f_T27_S9", cfg_full(27, seq2frame=True), True, True), ("full_T27_S7_notemb", cfg_full(27, with_time_emb=False), False, True),
        ("small_T81_S5_noclip", cfg_small(81), True, False), ("full_T243_S9", cfg_full(243), False, True),
        ("s2f_notemb_T27_S7", cfg_full(27, seq2frame=True, with_time_emb=False), False, True)]


@pytest.mark.parametrize("tag,cfg,traj,clip", DDIM, ids=[d[0] for d in DDIM])
def test_ddim_loop(tag, cfg, traj, clip):
    g = gold("ddim_" + tag)
    B, S = int(g["B"]), int(g["S"])
    sd = torch_sd(cfg, int(g["seed"]))
    inp = inputs(B, cfg.num_frame, int(g["input_seed"]))
    noise = inp["noise"][:, :1].contiguous() if cfg.seq2frame else inp["noise"]
    tabs = orc.diffusion_tables("cosine", 1000)
    out = orc.ddim_sample_loop(sd, tabs, inp["x2d"], noise, num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth,
                               clip_denoised=clip, seq2frame=cfg.seq2frame, return_trajectory=traj)
    if traj:
        assert np.abs(out[0].numpy() - g["y0"]).max() <= 5e-6
        assert np.abs(out[1].numpy() - g["x_reverse_diffusion"]).max() <= 5e-6
        assert np.abs(out[2].numpy() - g["x_start_est"]).max() <= 5e-6
        # the last-step quirk: final y equals the last (clamped) x_start estimate (DIFF:283-285)
        assert torch.equal(out[0], out[2][..., -1]) and torch.equal(out[0], out[1][..., -1])
    else:
        assert np.abs(out.numpy() - g["y0"]).max() <= 5e-6
    if clip:
        assert np.abs(g["y0"]).max() <= 1.0


def test_ddim_first_step_uses_alpha_not_sqrt_alpha():
    """DIFF:296 quirk: the implied-noise term is (y - alpha*x0)/sqrt(1-alpha), NOT (y - sqrt(alpha)*x0)/..."""
    g = gold("ddim_small_T81_S5")
    cfg = cfg_small(81)
    inp = inputs(int(g["B"]), 81, int(g["input_seed"]))
    tabs = orc.diffusion_tables("cosine", 1000)
    ac, so = tabs["alphas_cumprod"], tabs["sqrt_one_minus_alphas_cumprod"]
    x0 = torch.from_numpy(g["x_start_est"][..., 0])
    y = inp["noise"]
    t, tn = 999, 799
    quirk = x0 * ac[tn].sqrt() + (1 - ac[tn]).sqrt() * ((y - ac[t] * x0) / so[t])
    text = x0 * ac[tn].sqrt() + (1 - ac[tn]).sqrt() * ((y - ac[t].sqrt() * x0) / so[t])
    got = torch.from_numpy(g["x_reverse_diffusion"][..., 0])
    assert (quirk - got).abs().max() < 1e-6
    assert (text - got).abs().max() > 1e-6 or ac[t].sqrt() == ac[t]


def test_repeat_eta_and_plosses():
    g = gold("ddim_small_T27_S4_eta05_rep3")
    cfg = cfg_small(27)
    B, S, R = int(g["B"]), int(g["S"]), int(g["R"])
    sd = torch_sd(cfg, int(g["seed"]))
    inp = inputs(B * R, 27, int(g["input_seed"]))
    x2d, noise = inp["x2d"][:B], inp["noise"]
    step_noise = [hashed(f"eta_noise/{i}", tuple(noise.shape), 6) for i in range(S)]
    tabs = orc.diffusion_tables("cosine", 1000)
    o = orc.ddim_sample_loop(sd, tabs, x2d.repeat(R, 1, 1, 1), noise, num_timesteps=1000, sampling_timesteps=S,
                             depth=cfg.depth, eta=0.5, step_noise=step_noise)
    o = o.view(R, B, 27, 17, 3).mean(0)
    assert np.abs(o.numpy() - g["y0"]).max() <= 5e-6

    g = gold("plosses_small_T27")
    B = int(g["B"])
    sd = torch_sd(cfg, int(g["seed"]))
    inp = inputs(B, 27, int(g["input_seed"]))
    gt = inp["gt3d"] * float(g["gt_scale"])
    t = torch.from_numpy(g["t"]).long()
    assert np.abs(orc.q_sample(tabs, gt, t, inp["noise"]).numpy() - g["q_sample"]).max() <= TOL
    loss = orc.p_losses(sd, tabs, gt, inp["x2d"], t, inp["noise"], depth=cfg.depth, clip_loss=True)
    assert np.abs(loss.numpy() - g["loss"]).max() <= 5e-6


def test_evalmath():
    g = gold("evalmath")
    merged = orc.merge_flip_tta(torch.from_numpy(g["pred"]), torch.from_numpy(g["pred_flip"]), float(g["scale"]),
                                torch.from_numpy(g["target_mask"]), g["joints_left"].tolist(), g["joints_right"].tolist())
    assert np.array_equal(merged.numpy(), g["merged"])
    gt = torch.from_numpy(g["gt"]).view(-1, 17, 3)[torch.from_numpy(g["target_mask"]).view(-1)].unsqueeze(1)
    assert abs(orc.mpjpe(merged, gt).item() - float(g["mpjpe"])) < 1e-7


@pytest.mark.slow
@pytest.mark.skipif(not __import__("os").environ.get("D3D_SLOW_TESTS"),
                    reason="~1 min of CPU: set D3D_SLOW_TESTS=1 (the GPU suite checks the engine against this fixture anyway)")
def test_ddim_long_chain_T243_S50():
    g = gold("ddim_full_T243_S50")
    cfg = cfg_full(243)
    sd = torch_sd(cfg, int(g["seed"]))
    inp = inputs(1, 243, int(g["input_seed"]))
    tabs = orc.diffusion_tables("cosine", 1000)
    out = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], num_timesteps=1000, sampling_timesteps=50, depth=8)
    assert np.abs(out.numpy() - g["y0"]).max() <= 5e-6


def test_sequence_window_table():
    """chunk_index / gather_windows against the reference ChunkedGenerator's tables (captured in chunks.npz)."""
    g = gold("chunks")
    kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    for n, T in [(700, 243), (243, 243), (486, 243), (487, 243), (100, 27), (81, 27), (20, 27), (1, 9)]:
        tag = f"n{n}_T{T}"
        starts, mask = orc.chunk_index(n, T)
        assert np.array_equal(starts, g[tag + "/starts"]) and np.array_equal(mask, g[tag + "/mask"]), tag
        rng = np.random.RandomState(n * 1000 + T)
        p2 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32))
        w, _ = orc.gather_windows(p2, T)
        wf, _ = orc.gather_windows(p2, T, True, kl, kr)
        assert float(w.double().sum()) == float(g[tag + "/win_checksum"])
        assert float((wf.double() * torch.arange(1, 35, dtype=torch.float64).reshape(17, 2)).sum()) == float(g[tag + "/flip_checksum"])


@pytest.mark.parametrize("T,s2f", [(27, False), (243, False), (27, True)])
def test_trainedlike_family_denoise(T, s2f):
    """Second weight family (heavy-tailed weights, wide LayerNorm gains, O(1) position embeddings): the restatement reproduces the
    imported reference's outputs there too (bit-exact on the generating host); seq2seq and (round 4) seq2frame."""
    from diff3dhpe_amd.synth import synth_state_dict
    g = gold(f"denoise_trainedlike_{'s2f_' if s2f else ''}T{T}")
    cfg = cfg_full(T, seq2frame=s2f)
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, int(g["seed"]), family="trainedlike").items()}
    inp = inputs(2, T, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1)
    for key in [k for k in g.files if k.startswith("t") and k[1:].isdigit()]:
        o = orc.forward_denoise(sd, xcat, torch.full((2,), int(key[1:]), dtype=torch.long), depth=8, seq2frame=s2f)
        assert np.abs(o.numpy() - g[key]).max() <= TOL, key


def test_operand_rounding_emulation_is_off_by_default_and_scoped():
    """The bf16-operand emulation (the yardstick of the bf16 engine mode) changes results only inside its context, by the amount
    SURVEY Appendix B measured for bf16 operands (1e-3 .. 1e-1), and leaves the fp32 restatement bit-identical outside."""
    cfg = cfg_small(27)
    sd = torch_sd(cfg, 3)
    inp = inputs(2, 27, 9)
    xcat = torch.cat([inp["x2d"], inp["noise"]], dim=-1)
    t = torch.tensor([500, 20])
    a = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
    with orc.operand_rounding(torch.bfloat16):
        b = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
    c = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
    assert torch.equal(a, c)
    d = (a - b).abs().max().item()
    assert 1e-4 < d < 0.2, d


def test_pose_metrics_protocols():
    """evaluate()'s other three protocols (RUN:602-614): the oracle's restatement of LOSS:43-93, 132-142 against values of the reference's own
    p_mpjpe / n_mpjpe / mean_velocity_error (tests/golden/pose_metrics.npz, oracle/gen_golden.py gen_metrics) -- bit for bit, the nan of a
    one-frame batch included."""
    import numpy as np
    g = gold("pose_metrics")
    for tag in g["tags"]:
        pr, tg = torch.from_numpy(g[f"{tag}_pred"]).unsqueeze(1), torch.from_numpy(g[f"{tag}_gt"]).unsqueeze(1)
        n, s1, s2, s3, sv = orc.protocol_sums(pr, tg)
        ref = g[f"{tag}_ref"]
        for got, want in zip((s1 / n, s2 / n, s3 / n, sv / n), ref):
            assert got == want or (np.isnan(got) and np.isnan(want)), (tag, got, want)

