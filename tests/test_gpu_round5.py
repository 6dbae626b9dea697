"""Round-5 evidence on the MI355X, through the product API / the C ABI:
  * the F16X3 range guard is READ BY DEFAULT (no environment variable): precision "auto" -- the default -- repeats a flagged call on
    the exact-fp32 engine and matches the oracle, "f16x3" raises D3DError; forward_denoise, the sampling loops, p_losses and
    evaluate() (whose read rides on the batch's own synchronisation) are all covered,
  * the non-blocking ticket ABI underneath (d3d_engine_range_post / _take),
  * timestep indices outside the schedule tables: IndexError in Python, NaN rows + D3D_RANGE_INDEX at the C ABI."""
import ctypes as C
import os
import warnings

import numpy as np
import pytest
import torch

from helpers import cfg_full, inputs, maxabs, torch_sd
import diff3dhpe_amd as d3d
from diff3dhpe_amd import _lib
from diff3dhpe_amd.evaluate import evaluate
from diff3dhpe_amd.spec import DenoiserConfig

pytestmark = pytest.mark.gpu
GATE = 1e-4


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _outlier_channel(sd):
    """A checkpoint with ONE massive channel in the residual stream (|x| ~ 1e4 > 8188: outside the fp16 plane range); fp32 arithmetic
    handles it (row std ~ 440, no cancellation), so the exact-fp32 engine and the oracle agree to the usual distance."""
    sd["fusion_layer.bias"][7] += 1.0e4


def _all_channels(sd):
    """Every channel at ~1e4: out of range for F16X3 -- and ill-conditioned for ANY fp32 implementation (LayerNorm of 1e4 + O(1)), so
    this one is only compared with the product's own fp32 engine, bit for bit."""
    sd["fusion_layer.bias"] += 1.0e4


def _forty_sigma(sd):
    for k in ("Spatial_norm.bias", "Temporal_norm.bias"):
        sd[k] = sd[k] + 40.0 * sd[k.replace("bias", "weight")].abs().mean()


def _model(cfg, seed, mutate, precision=None, sampling=3):
    sd = torch_sd(cfg, seed)
    mutate(sd)
    net = d3d.HPE_model(d3d.S2F_NAME if cfg.seq2frame else d3d.S2S_NAME)(
        num_frame=cfg.num_frame, num_joints=17, in_chans=2, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=8, mlp_ratio=2.0,
        qkv_bias=True, qk_scale=None, drop_path_rate=0.1, with_time_emb=cfg.with_time_emb)
    net.load_state_dict(sd, strict=True)
    if precision is not None:
        net.precision = precision
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=sampling, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0, clipLoss=True).eval().cuda()
    return sd, net, diff


def test_default_precision_is_auto_and_the_guard_is_on_without_any_environment_variable():
    assert "D3D_CHECK_RANGE" not in os.environ
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=9, embed_dim=512, depth=1)
    assert net.precision == "auto" and net.range_check is True
    net.load_state_dict(torch_sd(DenoiserConfig(num_frame=9, embed_dim=512, depth=1), 3))
    eng = net.cuda().engine_for(_dev())
    assert eng.precision == "f16x3"                      # the engine "auto" starts on
    assert not net._engines_fb                           # the fp32 engine exists only once the guard has fired


@pytest.mark.parametrize("mutate,vs_oracle", [(_outlier_channel, True), (_forty_sigma, True), (_all_channels, False)],
                         ids=["outlier_channel", "forty_sigma", "all_channels"])
def test_auto_precision_repeats_flagged_calls_on_the_fp32_engine(mutate, vs_oracle):
    """The state dicts of test_f16x3_range_guard / test_range_guard_row_statistics_bit through the DEFAULT precision, no env var:
    forward_denoise, a whole sampling (with trajectories) and p_losses come out as the exact-fp32 engine computes them -- within
    1e-4 of the oracle where fp32 itself is well-conditioned -- with ONE warning; the same calls with precision "f16x3" raise."""
    from oracle import d3d_oracle as orc
    cfg = cfg_full(9)
    B, S = 3, 3
    inp = inputs(B, 9, 557)
    xcat = torch.cat([inp["x2d"], inp["noise"]], dim=-1)
    t = torch.tensor([999, 400, 3])
    gt = inp["gt3d"] * 0.5
    tabs = orc.diffusion_tables("cosine", 1000)

    sd, net, diff = _model(cfg, 77, mutate, sampling=S)                     # precision left at its default
    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter("always")
        out = net.forward_denoise(xcat.cuda(), t.cuda())
        _, y0, rev, x0s = diff(clean_3d_pose=torch.zeros(B, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                               init_noise=inp["noise"].cuda(), output_reverse_diffusion_3d=True)
        _, y0b = diff(clean_3d_pose=torch.zeros(B, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                      init_noise=inp["noise"].cuda())
        loss = diff.p_losses(gt.cuda(), inp["x2d"].cuda(), noise=inp["noise"].cuda(), t=t.cuda())
    ours = [w for w in wlog if "range guard fired" in str(w.message)]
    assert len(ours) == 1 and issubclass(ours[0].category, RuntimeWarning), [str(w.message) for w in wlog]
    g = net._guard
    assert g["flagged"] == 1 and g["reruns"] == 1 and net._on_fallback()     # first call flagged; the later ones went straight to fp32
    assert list(net._engines_fb) == [torch.cuda.current_device()]
    assert torch.equal(y0, y0b)

    _, net32, diff32 = _model(cfg, 77, mutate, precision="fp32", sampling=S)
    out32 = net32.forward_denoise(xcat.cuda(), t.cuda())
    _, y32, rev32, x0s32 = diff32(clean_3d_pose=torch.zeros(B, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                                  init_noise=inp["noise"].cuda(), output_reverse_diffusion_3d=True)
    loss32 = diff32.p_losses(gt.cuda(), inp["x2d"].cuda(), noise=inp["noise"].cuda(), t=t.cuda())
    for a, b in ((out, out32), (y0, y32), (rev, rev32), (x0s, x0s32), (loss, loss32)):
        assert torch.equal(a, b)                                             # "auto" after a flag IS the fp32 engine
    if vs_oracle:
        ref = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
        ref_y = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth)
        ref_l = orc.p_losses(sd, tabs, gt, inp["x2d"], t, inp["noise"], depth=cfg.depth, clip_loss=True)
        # the weighted loss is coef * (x0 - target)^2, coef <= 3: an error of the denoiser output arrives multiplied by 2 coef |x0 - target|,
        # so it is gated relative to the largest loss value (the outputs themselves: the absolute 1e-4 gate)
        lscale = max(1.0, float(ref_l.abs().max()))
        e = (maxabs(out, ref), maxabs(y0, ref_y), maxabs(loss, ref_l) / lscale)
        print(f"auto precision on a flagged checkpoint vs oracle: denoise {e[0]:.2e} sampling {e[1]:.2e} p_losses {e[2]:.2e} (relative to "
              f"the largest loss value {lscale:.1f})")
        assert max(e) <= GATE

    # the explicit F16X3 precision keeps raising -- by default, every entry point
    _, netx, diffx = _model(cfg, 77, mutate, precision="f16x3", sampling=S)
    with pytest.raises(_lib.D3DError, match="range"):
        netx.forward_denoise(xcat.cuda(), t.cuda())
    with pytest.raises(_lib.D3DError, match="range"):
        diffx(clean_3d_pose=torch.zeros(B, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False, init_noise=inp["noise"].cuda())
    with pytest.raises(_lib.D3DError, match="range"):
        diffx(clean_3d_pose=torch.zeros(B, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False, init_noise=inp["noise"].cuda(),
              output_reverse_diffusion_3d=True)
    with pytest.raises(_lib.D3DError, match="range"):
        diffx.p_losses(gt.cuda(), inp["x2d"].cuda(), noise=inp["noise"].cuda(), t=t.cuda())
    assert not netx._engines_fb


def test_auto_precision_stays_on_f16x3_for_a_healthy_checkpoint_and_returns_after_new_weights():
    cfg = cfg_full(9)
    inp = inputs(2, 9, 11)
    xcat = torch.cat([inp["x2d"], inp["noise"]], dim=-1).cuda()
    t = torch.tensor([500, 20]).cuda()
    sd, net, diff = _model(cfg, 5, lambda sd: None)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = net.forward_denoise(xcat, t)
        diff(clean_3d_pose=torch.zeros(2, 9, 17, 3).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False, init_noise=inp["noise"].cuda())
    assert net._guard["posted"] == 2 and net._guard["flagged"] == 0 and not net._engines_fb and not net._on_fallback()
    _, netx, _ = _model(cfg, 5, lambda sd: None, precision="f16x3")
    assert torch.equal(a, netx.forward_denoise(xcat, t))                     # "auto" on a healthy checkpoint IS the F16X3 engine
    # a flagged checkpoint moves the model to fp32; loading healthy weights brings it back
    bad = torch_sd(cfg, 5)
    _outlier_channel(bad)
    net.load_state_dict(bad)
    with pytest.warns(RuntimeWarning, match="range guard fired"):
        net.forward_denoise(xcat, t)
    assert net._on_fallback()
    net.load_state_dict(sd)
    assert not net._on_fallback()
    assert torch.equal(a, net.forward_denoise(xcat, t))
    # range_check = False: nothing is posted, nothing is read (the caller has taken the responsibility)
    net.range_check = False
    n = net._guard["posted"]
    net.forward_denoise(xcat, t)
    assert net._guard["posted"] == n


@pytest.mark.parametrize("precision", ["auto", "f16x3"])
def test_evaluate_reads_the_guard_at_its_own_synchronisation(precision):
    """evaluate() posts the tickets of its two samplings and reads them behind the batch's own synchronisation (the read-back of the
    two MPJPE sums): a flagged batch is repeated on the fp32 engine ("auto": MPJPE equal to the fp32 model's) or raises ("f16x3")."""
    cfg = cfg_full(9)
    B = 4
    inp = inputs(B, 9, 21)
    batch = {"inputs_2d": inp["x2d"].cuda(), "inputs_3d": inp["gt3d"].cuda(), "init_noise": inp["noise"].cuda(),
             "init_noise_flip": inp["noise"].cuda()}
    _, net, diff = _model(cfg, 9, _outlier_channel, precision=precision)
    if precision == "f16x3":
        with pytest.raises(_lib.D3DError, match="range"):
            evaluate(diff, [batch], verbose=False)
        return
    with pytest.warns(RuntimeWarning, match="range guard fired"):
        res = evaluate(diff, [batch, batch], verbose=False)
    # batch 1: two F16X3 tickets, repeated on the fp32 engine; batch 2: fp32 directly -- whose calls post tickets too since round 6
    # (the head fence's recompute bit is reported in every mode), read WITHOUT waiting: 2 + 2 + 2
    assert net._guard["reruns"] == 1 and net._guard["posted"] == 6
    _, _, diff32 = _model(cfg, 9, _outlier_channel, precision="fp32")
    ref = evaluate(diff32, [batch, batch], verbose=False)
    assert res["mpjpe_mm"] == ref["mpjpe_mm"] and res["frames"] == ref["frames"] == 2 * B * 9
    # a healthy model: tickets posted, none flagged, no fp32 engine
    _, neth, diffh = _model(cfg, 9, lambda sd: None)
    evaluate(diffh, [batch], verbose=False)
    assert neth._guard["posted"] == 2 and neth._guard["flagged"] == 0 and not neth._engines_fb


def test_range_ticket_abi():
    """d3d_engine_range_post / d3d_engine_range_take at the C ABI: a ticket covers the launches since the previous post, reading it
    waits for ITS event only, a poll never blocks, 256 tickets are kept."""
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=1)
    inp = inputs(2, 27, 3)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    _, net, diff = _model(cfg, 8, _all_channels, precision="f16x3", sampling=2)
    net.range_check = False
    _, neth, diffh = _model(cfg, 8, lambda sd: None, precision="f16x3", sampling=2)
    neth.range_check = False
    eb, eh = diff._engine(_dev()), diffh._engine(_dev())
    eb.range_flags(clear=True)
    t_empty = eb.post_range()
    assert eb.take_range(t_empty) == 0
    eb.ddim_sample(x2d, nz)
    eh.ddim_sample(x2d, nz)
    tb, th = eb.post_range(), eh.post_range()
    polled = eb.take_range(tb, block=False)                  # never blocks: None (not run yet) or the flags
    assert polled is None or polled & _lib.RANGE_ACT
    assert eb.take_range(tb) & _lib.RANGE_ACT and eh.take_range(th) == 0
    assert eb.take_range(tb) & _lib.RANGE_ACT                # a ticket can be read again ...
    assert eb.range_flags() == 0                             # ... and its snapshot has reset the engine's word
    t2 = eb.post_range()
    assert eb.take_range(t2) == 0                            # nothing launched since tb
    L = _lib.lib()
    f, r = C.c_uint32(0), C.c_int32(0)
    assert L.d3d_engine_range_take(eb._h, 10 ** 6, 1, C.byref(f), C.byref(r)) == _lib_code("EINVAL")
    assert L.d3d_engine_range_take(eb._h, -1, 1, C.byref(f), C.byref(r)) == _lib_code("EINVAL")
    for _ in range(256):
        last = eb.post_range()
    assert eb.take_range(last) == 0
    assert L.d3d_engine_range_take(eb._h, t2, 1, C.byref(f), C.byref(r)) == _lib_code("EINVAL") and b"expired" in L.d3d_last_error()
    # the sticky weight flag travels with every ticket
    _, netw, diffw = _model(cfg, 8, lambda sd: sd["STEblocks.0.mlp.fc1.weight"].__setitem__((3, 5), float("inf")), precision="f16x3", sampling=2)
    netw.range_check = False
    ew = diffw._engine(_dev())
    assert ew.take_range(ew.post_range()) & _lib.RANGE_WEIGHT


def _lib_code(name):
    return {"EINVAL": -1, "ESTATE": -2}[name]


def test_timestep_indices_outside_the_tables():
    """q_sample / the p_losses tail gather schedule tables by a caller-supplied timestep.  Python raises IndexError like
    the reference's table[t] (DIFF:21-24, 411); at the C ABI the row comes out NaN, no table entry is read and D3D_RANGE_INDEX rises."""
    cfg = DenoiserConfig(num_frame=9, embed_dim=512, depth=1)
    _, net, diff = _model(cfg, 4, lambda sd: None, sampling=2)
    inp = inputs(3, 9, 5)
    gt, x2d, nz = inp["gt3d"].cuda(), inp["x2d"].cuda(), inp["noise"].cuda()
    for bad in ([0, 1000, 5], [-1, 3, 5]):
        with pytest.raises(IndexError):
            diff.q_sample(gt, torch.tensor(bad).cuda(), nz)
        with pytest.raises(IndexError):
            diff.p_losses(gt, x2d, noise=nz, t=torch.tensor(bad).cuda())
    with pytest.raises(IndexError):
        diff.p_losses(gt, x2d, noise=nz, t=torch.tensor([1, 2]).cuda())      # a short t
    ok = diff.q_sample(gt, torch.tensor([0, 999, 5]).cuda(), nz)
    assert torch.isfinite(ok).all()
    eng = diff._engine(_dev())
    eng.range_flags(clear=True)
    t = torch.tensor([0, 10 ** 6, -7])
    out = eng.q_sample(gt, t, nz, check_t=False)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1:]).all()
    assert eng.range_flags() == _lib.RANGE_INDEX
    lo = eng.weighted_loss(gt, gt * 0.5, t, "l2", True, check_t=False)
    assert torch.isfinite(lo[0]).all() and torch.isnan(lo[1:]).all()
    assert eng.take_range(eng.post_range()) == _lib.RANGE_INDEX


# ------------------------------------------------------------------------------------------------ many ranks, ragged global batches
def _bench(args, env, timeout=900):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    return json.loads(lines[0])


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "D3D_FORCE_DIST",
                                                             "D3D_BENCH_ONE_DEVICE", "D3D_DIST_BACKEND")}
    env.update(extra)
    return env


SMALL = ["--steps", "1", "--warmup", "0", "--frames", "27", "--sampling", "3", "--no-cpu-baseline", "--no-extras", "--profile-steps", "0",
         "--no-selfcheck"]


@pytest.mark.parametrize("gb", [7, 4])
def test_bench_self_launch_five_ranks_on_one_device_ragged_global_batch(gb):
    """The unattended multi-GPU run, rehearsed as far as a one-GPU box allows (its process guard admits SIX processes on the card: this
    test process + FIVE ranks; the eight-process launch itself is exercised without a GPU in tests/test_compat_and_dist.py):
    `python bench.py --gpus 5 --global-batch 7` -> shards of 2,2,1,1,1 rows; `--global-batch 4` -> 1,1,1,1 and a rank with NO row that
    still joins the collective.  One JSON line, MPJPE equal to the one-rank value for the same global batch, start-up time and range
    flags reported over ranks, every rank building only its shard."""
    many = _bench(["--gpus", "5", "--global-batch", str(gb)] + SMALL, _env(D3D_BENCH_ONE_DEVICE="1", D3D_DIST_BACKEND="gloo"))
    one = _bench(["--gpus", "1", "--global-batch", str(gb)] + SMALL, _env())
    assert many["n_gpus"] == 5 and many["config"]["global_batch"] == gb == one["config"]["global_batch"]
    assert many["mpjpe_vs_synthetic_gt"] == one["mpjpe_vs_synthetic_gt"]
    r = many["ranks"]
    assert r["local_batch_min_max"] == [gb // 5, gb // 5 + 1] and r["range_flags"] == 0 and many["range_flags"] == 0 == one["range_flags"]
    assert 0 < r["startup_s_max_over_ranks"] < 600 and one["startup_s"] > 0
    assert many["dist"]["launcher"] == "self" and many["dist"]["world_size"] == 5
    print(f"5 ranks on one device, global batch {gb}: start-up (max over ranks) {r['startup_s_max_over_ranks']} s; one rank: {one['startup_s']} s")


def test_bench_strong_scaling_mode_names_itself():
    """--scaling strong fixes the GLOBAL batch (512 = BASELINE configs[2] unless --global-batch) and splits it over the ranks."""
    line = _bench(["--gpus", "1", "--scaling", "strong", "--global-batch", "6"] + SMALL, _env())
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 6 and "strong scaling" in line["config"]["workload"]
    weak = _bench(["--gpus", "1", "--batch", "6"] + SMALL, _env())
    assert weak["scaling"] == "weak" and weak["mpjpe_vs_synthetic_gt"] == line["mpjpe_vs_synthetic_gt"]


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: seq2frame evaluation
@pytest.mark.parametrize("n,T", [(100, 27), (27, 27), (5, 27), (1, 9), (40, 9)])
def test_seq2frame_window_gather_matches_the_oracle_bit_for_bit(n, T):
    """d3d_window_gather_s2f (one window per target frame, pad = (T - 1) / 2, edge replication; GEN:402-420, 492-512) against the oracle's
    restatement -- itself pinned on ChunkedGenerator_3dhp's output (chunks_s2f.npz, CPU suite) -- incl. the flipped copy and sub-ranges."""
    from oracle import d3d_oracle as orc
    from diff3dhpe_amd.engine import window_gather_s2f
    kl, kr = [5, 6, 7, 11, 12, 13], [2, 3, 4, 8, 9, 10]
    rng = np.random.RandomState(n * 977 + T)
    p2 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32))
    p3 = torch.from_numpy(rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32))
    w, _, _ = orc.gather_windows_s2f(p2, p3, None, T)
    wf, _, _ = orc.gather_windows_s2f(p2, p3, None, T, True, kl, kr)
    assert torch.equal(window_gather_s2f(p2.cuda(), T).cpu(), w)
    assert torch.equal(window_gather_s2f(p2.cuda(), T, True, kl, kr).cpu(), wf)
    if n >= 5:
        assert torch.equal(window_gather_s2f(p2.cuda(), T, first=2, count=3).cpu(), w[2:5])
    L = _lib.lib()
    out = torch.empty((n, T, 17, 2), device="cuda")
    z = (C.c_int32 * 0)()
    a = (C.c_void_p(p2.cuda().data_ptr()), n, T, 17, 2, 0, z, z, 0)
    assert L.d3d_window_gather_s2f(*a, 0, n + 1, C.c_void_p(out.data_ptr()), None) == -1          # range outside the sequence
    assert L.d3d_window_gather_s2f(C.c_void_p(p2.cuda().data_ptr()), n, T + 1, 17, 2, 0, z, z, 0, 0, n, C.c_void_p(out.data_ptr()), None) == -1


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "auto"])
def test_evaluate_seq2frame_3dhp_call_shape_against_the_reference(prec):
    """BASELINE configs[4] end to end: the 3DHP runner's evaluate() (run_..._3dhp.py:479-533) with an ...S2F... model WITHOUT time
    embedding (Experiments.sh:15-17), output_loss=True (forward()'s default there), flip-TTA, (B, 1, J, 3) targets and the frames'
    `valid` flags as mask -- on the DataLoader batches of sequence TS1 of the synthetic 3DHP-shaped data set, against the per-batch
    MPJPE the REFERENCE itself produced on those batches (evaluate_3dhp_s2f.npz; oracle == reference there)."""
    from conftest import gold
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.synth import synth_mocap_3dhp, hash_uniform
    g = gold("evaluate_3dhp_s2f")
    cfg = cfg_full(27, seq2frame=True, with_time_emb=False)
    S, bs = int(g["S"]), int(g["batch_size"])
    _, net, diff = _model(cfg, int(g["seed"]), lambda sd: None, precision=prec, sampling=S)
    test, train = synth_mocap_3dhp(0)
    ed = EvalData3DHP(test, ["TS1", "TS5"], 27, out_all=False, train_data=train)
    assert np.float32(ed.scale) == g["scale"]
    batches = []
    for bi, b in enumerate(ed.batches(bs, seq_filter="TS1")):
        B = b["inputs_2d"].shape[0]
        b["init_noise"] = torch.from_numpy(hash_uniform(f"eval3dhp/noise/{bi}", B * 17 * 3, 5).astype(np.float32).reshape(B, 1, 17, 3)) * 1.7
        b["init_noise_flip"] = torch.from_numpy(hash_uniform(f"eval3dhp/noise_flip/{bi}", B * 17 * 3, 5).astype(np.float32).reshape(B, 1, 17, 3)) * 1.7
        batches.append(b)
    kw = dict(scale=ed.scale, joints_left=ed.joints_left, joints_right=ed.joints_right, output_loss=True, unit_scale=1.0, verbose=False)
    worst = 0.0
    for bi, b in enumerate(batches):
        r = evaluate(diff, [b], **kw)
        assert r["frames"] == int(g["frames_per_batch"][bi])
        worst = max(worst, abs(r["mpjpe_mm"] - float(g["mpjpe_per_batch"][bi])))
    whole = evaluate(diff, batches, **kw)
    ref = float(np.dot(g["mpjpe_per_batch"], g["frames_per_batch"]) / g["frames_per_batch"].sum())
    print(f"3DHP seq2frame evaluate() [{prec}]: MPJPE {whole['mpjpe_mm']:.4f} mm (reference {ref:.4f}), worst batch deviation {worst:.2e} mm "
          f"at scale {ed.scale:.0f} mm")
    assert whole["frames"] == int(g["frames_per_batch"].sum())
    # 1e-4 on the normalised poses is 1e-4 * scale in mm; a mean of joint distances moves by no more than that
    assert worst <= GATE * ed.scale and abs(whole["mpjpe_mm"] - ref) <= GATE * ed.scale
    assert net._guard["flagged"] == 0


def test_evaluate_sequence_seq2frame_equals_the_batched_route():
    """evaluate_sequence() on a seq2frame model builds the per-frame windows, the flipped copy, the (n, 1, J, 3) targets and the
    `valid` mask on the device: same MPJPE as evaluate() over the host adaptor's DataLoader batches, bit for bit; a seq2seq model on
    the same 3DHP sequence takes the shifted-window table with `valid` ANDed in and agrees with ITS batched route."""
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.evaluate import evaluate_sequence
    from diff3dhpe_amd.synth import synth_mocap_3dhp, hash_uniform
    test, train = synth_mocap_3dhp(0)
    for s2f in (True, False):
        cfg = cfg_full(27, seq2frame=s2f, with_time_emb=False)
        _, net, diff = _model(cfg, 11, lambda sd: None, sampling=2)
        ed = EvalData3DHP(test, ["TS1", "TS5"], 27, out_all=not s2f, train_data=train)
        name, p2, p3, valid = ed.sequence("TS5")
        items = ed.num_items("TS5")
        To = 1 if s2f else 27
        nz = torch.from_numpy(hash_uniform("evalseq/nz", items * To * 17 * 3, 3).astype(np.float32).reshape(items, To, 17, 3))
        nzf = torch.from_numpy(hash_uniform("evalseq/nzf", items * To * 17 * 3, 3).astype(np.float32).reshape(items, To, 17, 3))
        kw = dict(scale=ed.scale, joints_left=ed.joints_left, joints_right=ed.joints_right, unit_scale=1.0)
        a = evaluate_sequence(diff, torch.from_numpy(p2), torch.from_numpy(p3), num_frames=27, batch_size=20, init_noise=nz.cuda(),
                              init_noise_flip=nzf.cuda(), valid=torch.from_numpy(valid), **kw)
        batches = []
        for bi, b in enumerate(ed.batches(20, seq_filter="TS5")):
            b["init_noise"], b["init_noise_flip"] = nz[20 * bi:20 * bi + 20], nzf[20 * bi:20 * bi + 20]
            batches.append(b)
        r = evaluate(diff, batches, verbose=False, **kw)
        assert a["frames"] == r["frames"] > 0 and a["mpjpe_mm"] == r["mpjpe_mm"], (s2f, a, r)


# ------------------------------------------------------------------------------------------------ the head kernel's run-time fence
@pytest.mark.parametrize("prec", ["f16x3", "fp32", "auto"])
def test_head_kernel_recompute_fence(prec):
    """With "head_fence" on (round 5's default; an option since round 6) k_head forms every row's three dot products TWICE from
    independently loaded weight fragments and compares them bit for bit (the one run-to-run deviation this library ever showed -- a wrong
    o[0] in ~1 launch of 60 beside a second process -- would show exactly there; round 6 identified it: a packed fp32 instruction form
    beside another wave's MFMAs, pinned out of every kernel at build time).  "head_inject" perturbs the FIRST evaluation of row 0:
    the kernel must repair the row by its third evaluation (results unchanged, whole sampling), raise D3D_RANGE_RECOMPUTE in EVERY
    precision, and the Python layer must report it once without raising or changing engines."""
    cfg = cfg_full(27)
    cfgd = type(cfg)(num_frame=27, embed_dim=512, depth=2)
    _, net, diff = _model(cfgd, 6, lambda sd: None, precision=prec, sampling=3)
    inp = inputs(3, 27, 41)
    z, x2d, nz = torch.zeros(3, 27, 17, 3).cuda(), inp["x2d"].cuda(), inp["noise"].cuda()
    eng = diff._engine(_dev())
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, clean = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
        assert eng.take_range(eng.post_range()) == 0
        raw = eng.ddim_sample(x2d, nz)
    eng.set_option("head_fence", 1)
    assert torch.equal(eng.ddim_sample(x2d, nz), raw)                             # the fence by itself changes nothing
    assert eng.take_range(eng.post_range()) == 0
    eng.set_option("head_inject", 1)
    try:
        poked = eng.ddim_sample(x2d, nz)
        assert eng.take_range(eng.post_range()) == _lib.RANGE_RECOMPUTE          # seen ...
        assert torch.equal(poked, raw)                                            # ... and repaired
        with pytest.warns(RuntimeWarning, match="two evaluations of a row disagreeing"):
            _, again = diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            net.flush_range_checks()     # (fp32: no precision flag can rise on that engine, so its guard is read WITHOUT waiting -- ADVICE
                                         #  r05 --: a fault bit shows at the next guarded call at the latest, or here)
        assert torch.equal(again, clean)
        g = net._guard
        assert g.get("recomputes") == 1 and g["flagged"] == 0 and g["reruns"] == 0 and not net._on_fallback()
        with warnings.catch_warnings():
            warnings.simplefilter("error")                                        # reported once per model
            diff(clean_3d_pose=z, noisy_2d_pose=x2d, output_loss=False, init_noise=nz)
            net.flush_range_checks()
        assert g.get("recomputes") == 2
    finally:
        eng.set_option("head_inject", 0)
        eng.set_option("head_fence", 0)
    assert eng.take_range(eng.post_range()) == 0
    # the single-op hook runs the same kernel (rows that are not a multiple of the 32-row workgroup included)
    X = torch.randn(77, 512, device="cuda")
    a = eng.head(X)
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_evaluate_seq2seq_3dhp_the_references_own_command_lines(prec):
    """Experiments.sh:15-17 as written: the ...S2S... model WITHOUT time embedding on MPI-INF-3DHP, out_all=True (27-frame windows, the last
    one shifted, its overlap mask ANDed with the frames' `valid` flags), the 3DHP runner's evaluate() call shape -- per test sequence
    (run_evaluation(): seq_filter), both routes (DataLoader batches of the host adaptor; evaluate_sequence on the device), against the
    per-batch MPJPE the REFERENCE produced on the synthetic 3DHP-shaped set (evaluate_3dhp_s2s.npz)."""
    from conftest import gold
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.evaluate import evaluate_sequence
    from diff3dhpe_amd.synth import synth_mocap_3dhp, hash_uniform
    g = gold("evaluate_3dhp_s2s")
    T, S, bs = 27, int(g["S"]), int(g["batch_size"])
    cfg = cfg_full(T, with_time_emb=False)
    _, net, diff = _model(cfg, int(g["seed"]), lambda sd: None, precision=prec, sampling=S)
    test, train = synth_mocap_3dhp(0)
    ed = EvalData3DHP(test, ["TS1", "TS5", "TS3"], T, out_all=True, train_data=train)
    kw = dict(scale=ed.scale, joints_left=ed.joints_left, joints_right=ed.joints_right, output_loss=True, unit_scale=1.0)
    for seq in ("TS1", "TS5", "TS3"):
        batches, nzs, nzfs = [], [], []
        for bi, b in enumerate(ed.batches(bs, seq_filter=seq)):
            B = b["inputs_2d"].shape[0]
            b["init_noise"] = torch.from_numpy(hash_uniform(f"eval3dhp_s2s/{seq}/noise/{bi}", B * T * 17 * 3, 5).astype(np.float32).reshape(B, T, 17, 3)) * 1.7
            b["init_noise_flip"] = torch.from_numpy(hash_uniform(f"eval3dhp_s2s/{seq}/noise_flip/{bi}", B * T * 17 * 3, 5).astype(np.float32).reshape(B, T, 17, 3)) * 1.7
            batches.append(b); nzs.append(b["init_noise"]); nzfs.append(b["init_noise_flip"])
        ref_e, ref_n = g[f"{seq}/mpjpe_per_batch"], g[f"{seq}/frames_per_batch"]
        for bi, b in enumerate(batches):
            r = evaluate(diff, [b], verbose=False, **kw)
            assert r["frames"] == int(ref_n[bi]) and abs(r["mpjpe_mm"] - float(ref_e[bi])) <= GATE * ed.scale, (seq, bi, r, ref_e[bi])
        whole = evaluate(diff, batches, verbose=False, **kw)
        ref = float(np.dot(ref_e, ref_n) / ref_n.sum())
        assert whole["frames"] == int(ref_n.sum()) and abs(whole["mpjpe_mm"] - ref) <= GATE * ed.scale
        _, p2, p3, valid = ed.sequence(seq)
        dev_route = evaluate_sequence(diff, torch.from_numpy(p2), torch.from_numpy(p3), num_frames=T, batch_size=bs, init_noise=torch.cat(nzs).cuda(),
                                      init_noise_flip=torch.cat(nzfs).cuda(), valid=torch.from_numpy(valid), **kw)
        assert dev_route["frames"] == whole["frames"] and dev_route["mpjpe_mm"] == whole["mpjpe_mm"]
        print(f"3DHP seq2seq evaluate() {seq} [{prec}]: MPJPE {whole['mpjpe_mm']:.4f} mm (reference {ref:.4f}), {whole['frames']} valid frames")


# ------------------------------------------------------------------------------------------------ bf16 operand mode (second-class)
@pytest.mark.parametrize("T,B", [(81, 25), (27, 80)])
def test_bf16_gemm_kernel_is_bit_identical_to_the_token_gemm_forms(T, B):
    """BF16 mode: qkv and fc1 on their own kernel (kernels_gemm_bf16q.hip: the hand-specialised two-phase k-loop of the fused F16X3 kernels
    with one bf16 MFMA per fragment pair) against the token GEMM's bf16 forms -- per element the same MFMAs in the same order and the same
    epilogue function, so a whole forward is bit-identical; M = B T 17 is not a multiple of 256 (the ragged last M-tile goes through the
    checked epilogue) and the workspace is filled with NaN first (operand pad rows are staged, never stored)."""
    cfg = type(cfg_full(T))(num_frame=T, embed_dim=512, depth=2)
    _, net, diff = _model(cfg, 21, lambda sd: None, precision="bf16", sampling=2)
    eng = diff._engine(_dev())
    inp = inputs(B, T, 77)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    assert (B * T * 17) % 256 != 0
    outs = []
    for on in (1, 0, 1):
        eng.set_option("bf16_gemm_kernel", on)
        eng._ws = None
        eng._workspace(B).view(torch.float32).fill_(float("nan"))
        outs.append(eng.ddim_sample(x2d, nz).clone())
    eng.set_option("bf16_gemm_kernel", 1)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
