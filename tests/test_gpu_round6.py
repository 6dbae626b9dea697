"""Round-6 evidence on the MI355X, through the product API / the C ABI:
  * fc2 + post-norm on the barrier-free k-loop (kernels_fc2_ring.hip: wave-private W slots, A through a four-slot ring with arrival
    counters in LDS) against the token GEMM's post-norm form, bit for bit."""
from functools import partial

import pytest
import torch

import diff3dhpe_amd as d3d
from conftest import gold
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict
from helpers import cfg_full, inputs, maxabs
from test_gpu_round4 import _product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,B,family,streams,delay", [(243, 33, "trainedlike", 1, 24), (243, 32, "uniform", 1, 0), (243, 64, "uniform", 2, 24),
                                                      (81, 97, "uniform", 1, 7)])
def test_fc2_ring_kernel_is_bit_identical_to_the_template_form(T, B, family, streams, delay):
    """"fc2_ring" (default): fc2 + post-norm on its own kernel without workgroup barriers in the k-loop (128 x 512 whole-row tiles, the
    template's own post-norm epilogue function, the ragged last tile by its checked form): bit for bit the token GEMM's post-norm form,
    whatever the start delay of waves 4-7.  Token counts that are and are not multiples of 128, one and two streams, NaN-filled
    workspace; no poll of an arrival counter times out (range-guard bit 16)."""
    cfg = cfg_full(T)
    _, diff = _product(cfg, 11 if family == "trainedlike" else 5, "f16x3", sampling=2, family=family)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng.set_option("streams", streams)
    inp = inputs(B, T, 82)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    eng.set_option("fc2_ring", 0)
    plain = eng.ddim_sample(x2d, nz).clone()
    eng.set_option("fc2_ring", 1)
    eng.set_option("fc2_ring_delay", delay)
    try:
        eng._workspace(B).view(torch.float32).fill_(float("nan"))
        eng.range_flags(clear=True)
        own = eng.ddim_sample(x2d, nz).clone()
        assert eng.range_flags() == 0
        assert torch.isfinite(own).all()
        assert torch.equal(own, plain)
    finally:
        eng.set_option("fc2_ring_delay", 24)


CTOR = {"nobias": (dict(qkv_bias=False), "uniform"), "qkscale": (dict(qk_scale=0.2), "uniform"),
        "eps": (dict(norm_layer=partial(torch.nn.LayerNorm, eps=1e-3)), "trainedlike")}


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag", list(CTOR))
def test_ctor_arguments_golden(tag, prec):
    """qkv_bias=False, qk_scale=0.2 and norm_layer=partial(nn.LayerNorm, eps=1e-3) (S2S:140-142, 184; the engine refused them until round
    6) against raw denoiser outputs of the imported reference (oracle/gen_golden.py gen_round6: T = 27, D = 512, depth 2; the oracle equal
    to the reference bit for bit while generating), in both gated precisions, range guard silent."""
    g = gold("denoise_ctor_args_T27")
    kw, family = CTOR[tag]
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=2)
    args = dict(num_frame=27, num_joints=17, in_chans=2, embed_dim=512, depth=2, num_heads=8, mlp_ratio=2., qkv_bias=True, qk_scale=None,
                drop_path_rate=0.1, with_time_emb=True)
    args.update(kw)
    net = d3d.HPE_model(d3d.S2S_NAME)(**args)
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, int(g["seed"]), family=family).items()}
    if tag == "nobias":
        sd = {k: v for k, v in sd.items() if not k.endswith("attn.qkv.bias")}
    net.load_state_dict(sd, strict=True)
    net.precision = prec
    net = net.cuda()
    inp = inputs(2, 27, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1).cuda()
    worst = 0.0
    for t in (999, 17):
        out = net.forward_denoise(xcat, torch.full((2,), t, dtype=torch.long, device="cuda"))
        worst = max(worst, maxabs(out, g[f"{tag}_t{t}"]))
    assert worst < 1e-4, (tag, prec, worst)
    eng = net.engine_for(torch.device("cuda", torch.cuda.current_device()))
    assert eng.range_flags() == 0


@pytest.mark.parametrize("B,T,prec", [(1, 243, "f16x3"), (2, 243, "f16x3"), (5, 243, "f16x3"), (3, 81, "f16x3"), (2, 27, "f16x3"), (3, 243, "bf16")])
def test_deep_operand_staging_of_the_small_batch_gemms_is_bit_identical(B, T, prec):
    """One-tile-per-workgroup GEMM launches (batches of a few sequences: proj on 128 x 128 tiles, fc1 / proj / qkv on 256 x 128) stage
    three / four k-tiles deep ("deep_stages", default on: a k-tile no longer lasts a DMA round trip) -- the same MFMAs in the same order
    as with two stages: the sampling is bit for bit the two-stage one, one and two streams, NaN-filled workspace (S2S:46-54, 84)."""
    cfg = cfg_full(T)
    _, diff = _product(cfg, 5, prec, sampling=2)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    inp = inputs(B, T, 84)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    try:
        eng.set_option("deep_stages", 0)
        plain = eng.ddim_sample(x2d, nz).clone()
        eng.set_option("deep_stages", 1)
        for streams in (2, 1):
            eng.set_option("streams", streams)
            eng._workspace(B).view(torch.float32).fill_(float("nan"))
            own = eng.ddim_sample(x2d, nz).clone()
            assert torch.isfinite(own).all()
            assert torch.equal(own, plain), streams
    finally:
        eng.set_option("deep_stages", 1)
        eng.set_option("streams", 2)


@pytest.mark.parametrize("M,N,K,epi", [(4131, 512, 512, "residual"), (300, 1024, 512, "gelu"), (8262, 1536, 512, "none"), (97, 512, 64, "none"),
                                       (1000, 512, 32, "residual")])
def test_deep_operand_staging_op_level(M, N, K, epi):
    """The same at the op level (d3d_op_linear), incl. K of one and two k-tiles (fewer k-tiles than stages) and ragged M."""
    from diff3dhpe_amd import _lib as L
    from diff3dhpe_amd import engine as E
    from helpers import hashed
    A = hashed(f"dsA{M}", (M, K), 41, 2.0).cuda()
    W = hashed(f"dsW{N}{K}", (N, K), 42, 1.0 / K ** 0.5).cuda()
    b = hashed("dsb", (N,), 43, 0.5).cuda()
    R = hashed(f"dsR{M}", (M, N), 44, 1.5).cuda() if epi == "residual" else None
    try:
        L.check(L.lib().d3d_engine_set_option(None, b"deep_stages", 0))
        y0 = E.op_linear(A, W, b, R, epi=epi, precision="f16x3")
        L.check(L.lib().d3d_engine_set_option(None, b"deep_stages", 1))
        y1 = E.op_linear(A, W, b, R, epi=epi, precision="f16x3")
    finally:
        L.check(L.lib().d3d_engine_set_option(None, b"deep_stages", 1))
    ref = A.double() @ W.double().t() + b.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "residual":
        ref = ref + R.double()
    assert maxabs(y1, ref.cpu()) < 3e-6 * (K / 32) ** 0.5 + 2e-6
    assert torch.equal(y0, y1)


def test_pose_metrics_kernel_against_the_reference_values():
    """d3d_pose_metrics (evaluate()'s P-MPJPE / N-MPJPE / MPJVE on the device, one thread per kept frame, fp64 inside, the 3 x 3 SVD by a
    Jacobi eigen-decomposition) against values of the reference's own functions (tests/golden/pose_metrics.npz): near / unrelated / mirrored
    (det R < 0) / planar (singular H) / mm-scale batches, the nan of a one-frame batch -- and the same batches with dropped frames spliced
    in (target_mask): the kept frames alone decide the values, the velocity differences run over consecutive KEPT frames (RUN:588-590)."""
    import numpy as np
    from diff3dhpe_amd.engine import pose_metrics
    g = gold("pose_metrics")
    rng = np.random.RandomState(3)
    for tag in g["tags"]:
        pr, tg, ref = g[f"{tag}_pred"], g[f"{tag}_gt"], g[f"{tag}_ref"]
        n = pr.shape[0]
        keep = np.ones(n + n // 3 + 2, dtype=bool)
        keep[rng.choice(len(keep), len(keep) - n, replace=False)] = False
        prm = rng.uniform(-5, 5, (len(keep), 17, 3)).astype(np.float32)
        tgm = rng.uniform(-5, 5, (len(keep), 17, 3)).astype(np.float32)
        prm[keep], tgm[keep] = pr, tg
        for P, G, M in ((pr, tg, None), (prm, tgm, keep)):
            kept, en, ep, ev = pose_metrics(torch.from_numpy(P).cuda(), torch.from_numpy(G).cuda(), None if M is None else torch.from_numpy(M).cuda())
            assert kept == n
            for name, got, want in (("p_mpjpe", ep, ref[1]), ("n_mpjpe", en, ref[2]), ("mpjve", ev, ref[3])):
                if np.isnan(want):
                    assert np.isnan(got), (tag, name, got)
                else:
                    assert abs(got - want) <= 2e-6 * abs(want) + 1e-7, (tag, name, got, want)     # (the reference computes these in fp32)


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_evaluate_returns_the_four_protocols(prec):
    """evaluate() end to end (two DDIM samplings per window, merge, mask, the four running sums weighted by each batch's kept frames,
    RUN:562-614) against the oracle's sampler + its restatement of the reference's metric functions: two batches, one with masked frames,
    through as_reference_tuple() -- the reference's own return shape (e1, e2, e3, ev, N, epoch_time)."""
    from diff3dhpe_amd.evaluate import evaluate, as_reference_tuple, flip_2d, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    from oracle import d3d_oracle as orc
    from helpers import cfg_small, hashed, torch_sd, build_product
    cfg = cfg_small(27)
    _, diff = build_product(cfg, 12, sampling=3, precision=prec)
    sd, tabs = torch_sd(cfg, 12), orc.diffusion_tables("cosine", 1000)
    kw = dict(num_timesteps=1000, sampling_timesteps=3, depth=cfg.depth)
    batches, tot, N = [], [0.0, 0.0, 0.0, 0.0], 0
    for b, (B, seed) in enumerate(((3, 101), (2, 202))):
        inp = inputs(B, 27, seed)
        mask = torch.ones(B, 27, dtype=torch.bool)
        if b == 0:
            mask[1, :11] = False
        nz_f = hashed(f"flipnoise{b}", tuple(inp["noise"].shape), 1)
        batches.append({"inputs_2d": inp["x2d"], "inputs_3d": inp["gt3d"], "target_mask": mask, "init_noise": inp["noise"], "init_noise_flip": nz_f})
        p = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], **kw)
        pf = orc.ddim_sample_loop(sd, tabs, flip_2d(inp["x2d"], H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT), nz_f, **kw)
        merged = orc.merge_flip_tta(p, pf, 1.3, mask)
        gtm = inp["gt3d"].view(-1, 17, 3)[mask.view(-1)].unsqueeze(1)
        n, s1, s2, s3, sv = orc.protocol_sums(merged, gtm)
        N += n
        for i, v in enumerate((s1, s2, s3, sv)):
            tot[i] += v
    e1, e2, e3, ev, frames, secs = as_reference_tuple(evaluate(diff, batches, scale=1.3, verbose=False))
    assert frames == N and secs > 0
    for name, got, want in (("mpjpe", e1, tot[0]), ("p_mpjpe", e2, tot[1]), ("n_mpjpe", e3, tot[2]), ("mpjve", ev, tot[3])):
        assert abs(got - want / N * 1000) < 0.05, (name, got, want / N * 1000)      # mm at scale 1.3: 1e-4 of the sampler's gate and to spare
    only1 = evaluate(diff, batches, scale=1.3, verbose=False, all_protocols=False)
    assert only1["p_mpjpe_mm"] is None and abs(only1["mpjpe_mm"] - e1) < 1e-9


def test_run_evaluation_action_wise_averages():
    """run_evaluation() (RUN:712-766): one evaluate() per action name over the actions with that prefix, action-wise means of the four
    protocols, totals -- against the same evaluate() calls made by hand with the generator in the same state."""
    from diff3dhpe_amd.data import EvalData, MocapMeta
    from diff3dhpe_amd.evaluate import evaluate, run_evaluation, as_reference_tuple
    from diff3dhpe_amd.synth import synth_mocap, SYNTH_JOINTS_LEFT as JL, SYNTH_JOINTS_RIGHT as JR
    from helpers import cfg_small, build_product
    pos, cams, kp, meta = synth_mocap(0)
    ed = EvalData(MocapMeta(pos, cams, JL, JR), kp, meta["keypoints_symmetry"], ["S9", "S11"], 27)
    _, diff = build_product(cfg_small(27), 31, sampling=2, precision="f16x3")
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    res = run_evaluation(diff, ed, batch_size=4, verbose=False)
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    by_hand = {}
    for name in ed.action_names():
        by_hand[name] = as_reference_tuple(evaluate(diff, ed.batches(4, action_filter=[name]), scale=ed.scale, joints_left=ed.joints_left,
                                                    joints_right=ed.joints_right, verbose=False))
    assert list(res["actions"]) == ["Walk", "Sit", "Eat", "Wait"]
    for name, t in by_hand.items():
        assert res["actions"][name][:5] == t[:5], name
    for i, key in enumerate(("mpjpe_mm", "p_mpjpe_mm", "n_mpjpe_mm", "mpjve_mm")):
        assert abs(res[key] - sum(t[i] for t in by_hand.values()) / 4) < 1e-9
    assert res["frames"] == sum(t[4] for t in by_hand.values()) == sum(int(it["target_mask"].sum()) for it in ed.items())
    only = run_evaluation(diff, ed, batch_size=4, action_filter=["Wa"], verbose=False)
    assert list(only["actions"]) == ["Walk", "Wait"]


def test_run_evaluation_3dhp_per_sequence_and_stored_predictions(tmp_path):
    """run_evaluation_3dhp() (run_..._3dhp.py:593-632) with a seq2frame model on the synthetic 3DHP-shaped data set: one evaluate() per test
    sequence with forward()'s default output_loss, sequence-wise averages of the four protocols, and data_inference[seq] = the kept frames'
    merged predictions as (3, J, N) -- equal to what evaluate() computes by hand with the generator in the same state; inference_data.mat
    readable by scipy."""
    import numpy as np
    import scipy.io as scio
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.evaluate import evaluate, run_evaluation_3dhp, as_reference_tuple
    from diff3dhpe_amd.synth import synth_mocap_3dhp
    from helpers import cfg_small
    test, train = synth_mocap_3dhp(0)
    ed = EvalData3DHP(test, ["TS1", "TS5"], 27, out_all=False, train_data=train)
    _, diff = _product(cfg_small(27, seq2frame=True, with_time_emb=False), 9, "f16x3", sampling=2)
    mat = str(tmp_path / "inference_data.mat")
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    res = run_evaluation_3dhp(diff, ed, batch_size=32, verbose=False, mat_path=mat)
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    for name in ("TS1", "TS5"):
        r = evaluate(diff, ed.batches(32, seq_filter=name), scale=ed.scale, joints_left=ed.joints_left, joints_right=ed.joints_right,
                     verbose=False, unit_scale=1.0, output_loss=True, collect_predictions=True)
        assert res["sequences"][name][:5] == as_reference_tuple(r)[:5]
        valid = int(sum(int(it["target_mask"].sum()) for it in ed.items(seq_filter=name)))
        assert res["data_inference"][name].shape == (3, 17, valid) == tuple(r["predictions"].permute(2, 1, 0).shape)
        assert np.array_equal(res["data_inference"][name], r["predictions"].permute(2, 1, 0).numpy())
    for i, key in enumerate(("mpjpe_mm", "p_mpjpe_mm", "n_mpjpe_mm", "mpjve_mm")):
        assert abs(res[key] - (res["sequences"]["TS1"][i] + res["sequences"]["TS5"][i]) / 2) < 1e-9 and np.isfinite(res[key])
    m = scio.loadmat(mat)
    assert np.array_equal(m["TS1"], res["data_inference"]["TS1"]) and m["TS5"].shape[:2] == (3, 17)

