"""Round-3 parity evidence on the MI355X, through the product API / the C ABI:
  * a second, "trained-like" weight family (goldens made by the imported reference: oracle/gen_golden.py::gen_trainedlike),
  * the row-statistics bit of the F16X3 range guard,
  * the SURVEY section 8(f) rows (eta > 0, repeat_n, p_losses) at the bench width D = 512 in the DEFAULT precision,
  * the reference runner's own multi-device entry (thop-style hooks, nn.DataParallel wrap, module.-prefixed checkpoint),
  * the "streams" engine option (two half-batches on two HIP streams): bit-identical to one stream, eager and graph."""
import os

import numpy as np
import pytest
import torch

from conftest import gold
from helpers import cfg_full, cfg_small, inputs, hashed, build_product, maxabs, torch_sd
import diff3dhpe_amd as d3d
from diff3dhpe_amd import _lib
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
GATE = 1e-4
PRECS = ["fp32", "f16x3"]


def _tl_sd(cfg, seed):
    return {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed, family="trainedlike").items()}


def _tl_product(cfg, seed, prec, sampling=9):
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=cfg.num_frame, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8,
                                      mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_path_rate=0.1)
    net.load_state_dict(_tl_sd(cfg, seed), strict=True)
    net.precision = prec
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=sampling, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0, clipLoss=True).eval().cuda()
    return net, diff


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("T", [27, 243])
def test_trainedlike_family_denoise_golden(T, prec):
    """Heavy-tailed weights (|w| up to 8), LayerNorm gains in [0.05, 6] with biases up to 3 gamma, O(1) position embeddings, sharp
    softmax logits (up to ~40): either precision passes the 1e-4 gate against the REFERENCE's output, and the F16X3 range guard
    stays silent (if it ever fires here, the result must not be trusted -- and the test says so instead of passing)."""
    g = gold(f"denoise_trainedlike_T{T}")
    cfg = cfg_full(T)
    net, diff = _tl_product(cfg, int(g["seed"]), prec)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng.range_flags(clear=True)
    inp = inputs(2, T, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1).cuda()
    worst = 0.0
    for key in [k for k in g.files if k.startswith("t") and k[1:].isdigit()]:
        out = net.forward_denoise(xcat, torch.full((2,), int(key[1:]), dtype=torch.long, device="cuda"))
        worst = max(worst, maxabs(out, g[key]))
    if "tmixed" in g.files:
        worst = max(worst, maxabs(net.forward_denoise(xcat, torch.from_numpy(g["tmixed_t"]).long().cuda()), g["tmixed"]))
    flags = eng.range_flags()
    print(f"trained-like denoise T={T} [{prec}]: max-abs {worst:.3e}, range flags {flags}")
    assert flags == 0, "range guard fired on the trained-like family: use precision='fp32' for such a checkpoint"
    assert worst <= GATE


@pytest.mark.parametrize("prec", PRECS)
def test_trainedlike_family_ddim_golden(prec):
    g = gold("ddim_trainedlike_T81_S9")
    cfg = cfg_full(81)
    _, diff = _tl_product(cfg, int(g["seed"]), prec, sampling=int(g["S"]))
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng.range_flags(clear=True)
    inp = inputs(2, 81, int(g["input_seed"]))
    _, y0, rev, x0s = diff(torch.zeros_like(inp["noise"]).cuda(), inp["x2d"].cuda(), None, True, False, init_noise=inp["noise"].cuda())
    e = max(maxabs(y0, g["y0"]), maxabs(x0s, g["x_start_est"]))
    print(f"trained-like ddim T=81 S=9 [{prec}]: max-abs {e:.3e}")
    assert eng.range_flags() == 0
    assert e <= GATE


@pytest.mark.parametrize("off,expect", [(2.0, False), (8.0, False), (40.0, True)])
def test_range_guard_row_statistics_bit(off, expect):
    """D3D_RANGE_STATS: the folded LayerNorm works from one-pass row statistics, whose error grows like eps (1 + mean^2 / var).
    Post-norm biases pushed to `off` standard deviations of the row: silent at 2 and 8 sigma (where the parity test
    test_folded_layernorm_statistics_with_offset_rows shows 2.5e-6 / 1.5e-5 against the oracle), raised at 40 sigma -- before the
    1e-4 gate is at risk (~25 sigma) a caller that checks the guard is told to use precision='fp32'."""
    cfg = cfg_full(9)
    sd = torch_sd(cfg, 77)
    for k in ("Spatial_norm.bias", "Temporal_norm.bias"):
        sd[k] = sd[k] + off * sd[k.replace("bias", "weight")].abs().mean()
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=9, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8, mlp_ratio=2.0,
                                      qkv_bias=True, qk_scale=None, drop_path_rate=0.1)
    net.load_state_dict(sd, strict=True)
    net.precision = "f16x3"
    net.range_check = False              # the raw word is read below (the default reading: tests/test_gpu_round5.py)
    net = net.cuda()
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = net.engine_for(dev)
    eng.range_flags(clear=True)
    inp = inputs(3, 9, 557)
    out = net.forward_denoise(torch.cat([inp["x2d"], inp["noise"]], dim=-1).cuda(), torch.tensor([999, 400, 3]).cuda())
    assert torch.isfinite(out).all()
    f = eng.range_flags()
    assert bool(f & _lib.RANGE_STATS) == expect, (off, f)
    assert not (f & (_lib.RANGE_ACT | _lib.RANGE_WEIGHT))
    if expect:
        net.forward_denoise(torch.cat([inp["x2d"], inp["noise"]], dim=-1).cuda(), torch.tensor([999, 400, 3]).cuda())
        with pytest.raises(_lib.D3DError, match="standard deviations"):
            eng.check_range()
        net.precision = "fp32"                       # the remedy: two-pass row kernels, no folded statistics, no flag
        e32 = net.engine_for(dev)
        e32.range_flags(clear=True)
        net.forward_denoise(torch.cat([inp["x2d"], inp["noise"]], dim=-1).cuda(), torch.tensor([999, 400, 3]).cuda())
        assert e32.range_flags() == 0


@pytest.mark.parametrize("prec", PRECS)
def test_eta_and_repeat_n_at_bench_width(prec):
    """SURVEY 8(f) row 3 where the product default runs it: D = 512 (k_head's sigma * noise term, the per-step noise of a
    repeat_n = 2 batch), eta = 0.5, against the CPU oracle (DIFF:290-297, 433-448)."""
    from oracle import d3d_oracle as orc
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=8)
    B, S, R = 2, 4, 2
    _, diff = build_product(cfg, 61, sampling=S, eta=0.5, precision=prec)
    inp = inputs(B * R, 27, 301)
    step_noise = torch.stack([hashed(f"eta512/{i}", tuple(inp["noise"].shape), 7) for i in range(S)])
    _, y0 = diff(clean_3d_pose=torch.zeros(B, 27, 17, 3).cuda(), noisy_2d_pose=inp["x2d"][:B].cuda(), output_loss=False,
                 repeat_n=R, init_noise=inp["noise"].cuda(), step_noise=step_noise.cuda())
    ref = orc.ddim_sample_loop(torch_sd(cfg, 61), orc.diffusion_tables("cosine", 1000), inp["x2d"][:B].repeat(R, 1, 1, 1), inp["noise"],
                               num_timesteps=1000, sampling_timesteps=S, depth=8, eta=0.5, step_noise=list(step_noise))
    ref = ref.view(R, B, 27, 17, 3).mean(0)
    e = maxabs(y0, ref)
    print(f"eta=0.5 repeat_n=2 D=512 [{prec}]: max-abs {e:.3e}")
    assert y0.shape == (B, 27, 17, 3) and e <= GATE


@pytest.mark.parametrize("prec", PRECS)
def test_p_losses_at_bench_width_without_time_embedding(prec):
    """SURVEY 8(f) row 2 as RUN3DHP's evaluate() uses it (..._3dhp.py:517-520: with_time_emb = False, output_loss = True):
    q_sample + per-row timesteps + denoiser + weighted loss at D = 512, against the CPU oracle."""
    from oracle import d3d_oracle as orc
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=8, with_time_emb=False)
    B = 3
    _, diff = build_product(cfg, 62, sampling=5, precision=prec)
    inp = inputs(B, 27, 401)
    gt = (inp["gt3d"] * 0.5)
    t = torch.tensor([999, 12, 500], dtype=torch.long)
    loss = diff.p_losses(gt.cuda(), inp["x2d"].cuda(), noise=inp["noise"].cuda(), t=t.cuda())
    tabs = orc.diffusion_tables("cosine", 1000)
    ref = orc.p_losses(torch_sd(cfg, 62), tabs, gt, inp["x2d"], t, inp["noise"], depth=8, clip_loss=True)
    e = maxabs(loss, ref)
    print(f"p_losses D=512 no time embedding [{prec}]: max-abs {e:.3e}")
    assert e <= GATE
    # the 4-tuple / loss-carrying forward of the eval branch (DIFF:433-448 with output_loss=True)
    torch.manual_seed(5)
    lp, pred = diff(clean_3d_pose=gt.cuda(), noisy_2d_pose=inp["x2d"].cuda(), noise=inp["noise"].cuda(), output_loss=True,
                    init_noise=inp["noise"].cuda())
    assert lp.shape == gt.shape and pred.shape == gt.shape and torch.isfinite(lp).all()


def test_reference_runner_multi_device_entry_on_one_device(tmp_path):
    """What the unchanged runner does around its model (RUN:191-235, 577-582), executed: forward hooks on every sub-module and a
    positional call with CPU tensors (thop.profile, RUN:196), then nn.DataParallel(model).cuda(), a module.-prefixed checkpoint
    loaded the reference's way, and two keyword calls (flip-TTA pair) -- outputs bit-equal to the bare module's.  The parameter
    holders never run a forward, so the hooks never fire (thop therefore reports ~0 MACs: INTEGRATION.md says so)."""
    from diff3dhpe_amd.checkpoint import load_checkpoint, save_checkpoint
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=2)
    mk = lambda: d3d.GaussianDiffusion(model=d3d.HPE_model(d3d.S2S_NAME)(num_frame=27, embed_dim=512, depth=2, num_heads=8, mlp_ratio=2.0,
                                                                         qkv_bias=True, qk_scale=None, drop_path_rate=0.1),
                                       timesteps=1000, sampling_timesteps=3, loss_type="l2", clip_denoised=True, beta_schedule="cosine",
                                       ddim_sampling_eta=0.0, clipLoss=True)
    src = mk()
    src.model.load_state_dict(torch_sd(cfg, 33))
    path = str(tmp_path / "epoch_3.bin")
    save_checkpoint(src, path, epoch=3)                           # keys carry the DataParallel 'module.' prefix (RUN:459)
    inp = inputs(2, 27, 71)
    x2d, nz, gt = inp["x2d"], inp["noise"], inp["gt3d"]

    diff = mk().eval()
    fired = []
    hooks = [m.register_forward_hook(lambda mod, i, o: fired.append(type(mod).__name__)) for m in diff.modules()]
    torch.manual_seed(3)
    res = diff(gt, x2d, None, False, False)                       # RUN:196: CPU tensors, positional
    assert res[0] is None and res[1].device.type == "cpu" and res[1].shape == gt.shape
    assert set(fired) <= {"GaussianDiffusion"}, fired             # no holder forward ever runs
    for h in hooks:
        h.remove()

    wrapped = torch.nn.DataParallel(diff).cuda()                  # RUN:216-218
    meta = load_checkpoint(wrapped, path)                         # RUN:226-235
    assert meta["epoch"] == 3 and not meta["unexpected_keys"]
    bare = src.eval().cuda()
    outs = []
    for x in (x2d, -x2d):                                         # RUN:577-582: two keyword calls per batch
        _, a = wrapped(clean_3d_pose=gt.cuda(), noisy_2d_pose=x.cuda(), output_loss=False, init_noise=nz.cuda())
        _, b = bare(clean_3d_pose=gt.cuda(), noisy_2d_pose=x.cuda(), output_loss=False, init_noise=nz.cuda())
        assert torch.equal(a, b)
        outs.append(a)
    assert not torch.equal(outs[0], outs[1])
    # a replica as DataParallel builds them for several devices (plain-tensor parameters): shares the source's engine slot,
    # uploads nothing new, same bits
    rep = torch.nn.parallel.replicate(wrapped.module, [torch.cuda.current_device()])[0]
    _, c = rep(clean_3d_pose=gt.cuda(), noisy_2d_pose=x2d.cuda(), output_loss=False, init_noise=nz.cuda())
    assert torch.equal(c, outs[0])
    assert rep.model._engines is wrapped.module.model._engines and len(rep.model._engines) == 1
    # a second device in the same process is refused loudly (one process per GPU is the supported form)
    with pytest.raises(_lib.D3DError, match="ONE device per process"):
        wrapped.module.model.engine_for(torch.device("cuda", torch.cuda.current_device() + 1))


@pytest.mark.parametrize("T,B", [(27, 5), (243, 3)])
def test_two_stream_option_is_bit_identical(T, B):
    """"streams" = 2 (d3d_engine_set_option): the batch as two half-batches on two HIP streams -- eager and under hipGraph replay,
    with trajectory capture and with eta > 0 step noise (whose per-step stride is the WHOLE batch's) -- gives the bits of the
    one-stream call; odd batches split (B + 1) / 2 + B / 2; B = 1 cannot split and still works."""
    cfg = cfg_full(T)
    S = 3
    _, diff = build_product(cfg, 9, sampling=S, precision="f16x3")
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = diff._engine(dev)
    inp = inputs(B, T, 88)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    eng.set_option("streams", 1)
    ref, rrev, rx0 = eng.ddim_sample(x2d, nz, trajectory=True)
    ref = ref.clone()
    eng.set_option("streams", 2)
    out, rev, x0 = eng.ddim_sample(x2d, nz, trajectory=True)
    assert torch.equal(out, ref) and torch.equal(rev, rrev) and torch.equal(x0, rx0)
    assert torch.equal(eng.ddim_sample(x2d, nz), ref)
    eng.set_graph_mode(True)
    try:
        for _ in range(2):
            assert torch.equal(eng.ddim_sample(x2d, nz), ref)
    finally:
        eng.set_graph_mode(False)
    one = eng.ddim_sample(x2d[:1].contiguous(), nz[:1].contiguous())
    assert torch.equal(one, ref[:1])
    eng.set_option("streams", 1)
    if T == 27:   # eta > 0: supplied per-step noise (S, B, ...) is indexed with the whole batch's stride by both halves
        _, d2 = build_product(cfg, 9, sampling=S, eta=0.7, precision="f16x3")
        e2 = d2._engine(dev)
        sn = torch.stack([hashed(f"ts/{i}", tuple(nz.shape), 3) for i in range(S)]).cuda()
        a = e2.ddim_sample(x2d, nz, sn).clone()
        e2.set_option("streams", 2)
        assert torch.equal(e2.ddim_sample(x2d, nz, sn), a)
        e2.set_option("streams", 1)


def test_set_option_rejects_what_it_does_not_know():
    cfg = cfg_small(9)
    _, diff = build_product(cfg, 1, sampling=2)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    with pytest.raises(_lib.D3DError, match="unknown option"):
        eng.set_option("no_such_switch", 1)
    with pytest.raises(_lib.D3DError, match="streams"):
        eng.set_option("streams", 3)
    # the alternative F16X3 flows the options select stay inside the gate (and differ only in rounding)
    from oracle import d3d_oracle as orc
    cfg = DenoiserConfig(num_frame=9, embed_dim=512, depth=2)
    _, diff = build_product(cfg, 4, sampling=2, precision="f16x3")
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    inp = inputs(2, 9, 5)
    ref = orc.ddim_sample_loop(torch_sd(cfg, 4), orc.diffusion_tables("cosine", 1000), inp["x2d"], inp["noise"], num_timesteps=1000,
                               sampling_timesteps=2, depth=2)
    for key in (None, "fused_postnorm", "fold_layernorm"):
        if key:
            eng.set_option(key, 0)
        assert maxabs(eng.ddim_sample(inp["x2d"].cuda(), inp["noise"].cuda()), ref) <= GATE, key
        if key:
            eng.set_option(key, 1)


def test_trainedlike_family_on_the_large_batch_kernels():
    """The per-matrix weight scale (2^k, k < 12 for the LayerNorm-folded weights of this family) in the kernels the goldens do
    not reach: B = 32 at T = 243 runs the persistent 256x256 walk, its tail slices and the whole-row fc2 form with k = 9 .. 11;
    the result must be bit for bit the one of ragged small chunks (256x128 tiles, one tile per workgroup), with the guard silent."""
    cfg = cfg_full(243)
    _, diff = _tl_product(cfg, 11, "f16x3", sampling=1)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng.range_flags(clear=True)
    inp = inputs(32, 243, 500)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    big = eng.ddim_sample(x2d, nz).clone()
    for lo in (0, 13, 26):
        hi = min(lo + 13, 32)
        assert torch.equal(eng.ddim_sample(x2d[lo:hi].contiguous(), nz[lo:hi].contiguous()), big[lo:hi])
    assert eng.range_flags() == 0 and torch.isfinite(big).all() and big.abs().max().item() <= 1.0


@pytest.mark.parametrize("prec", PRECS)
def test_frame_counts_outside_the_reference_configs(prec):
    """num_frame is free in the reference; its configs use 27 / 81 / 243.  Other counts take other kernel instantiations:
    T = 300 is beyond the 256-frame fp16-MFMA temporal attention (F16X3 then runs the row-kernel flow with the generic attention
    for the temporal blocks), T = 100 / 200 take the 4- and 7-key-tile forms, T = 1 is a single-frame 'video'.  Bench width for
    the long ones (one denoiser evaluation, B = 1), the small model for a whole sampling."""
    from oracle import d3d_oracle as orc
    for T in (300, 200, 100):
        cfg = cfg_full(T)
        net, _ = build_product(cfg, 31, sampling=2, precision=prec)
        inp = inputs(1, T, 310 + T)
        xcat = torch.cat([inp["x2d"], inp["noise"]], dim=-1)
        t = torch.tensor([611])
        out = net.forward_denoise(xcat.cuda(), t.cuda())
        ref = orc.forward_denoise(torch_sd(cfg, 31), xcat, t, depth=cfg.depth)
        assert maxabs(out, ref) <= GATE, (T, maxabs(out, ref))
    for T in (1, 300):
        cfg = cfg_small(T)
        _, diff = build_product(cfg, 32, sampling=3, precision=prec)
        inp = inputs(2, T, 320 + T)
        _, y0 = diff(clean_3d_pose=torch.zeros_like(inp["noise"]).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                     init_noise=inp["noise"].cuda())
        ref = orc.ddim_sample_loop(torch_sd(cfg, 32), orc.diffusion_tables("cosine", 1000), inp["x2d"], inp["noise"],
                                   num_timesteps=1000, sampling_timesteps=3, depth=cfg.depth)
        assert maxabs(y0, ref) <= GATE, (T, maxabs(y0, ref))


@pytest.mark.parametrize("prec", PRECS)
def test_joint_counts_other_than_17(prec):
    """num_joints is a constructor argument of the reference models (17 in every shipped config).  21 joints stay on the
    wave-private spatial attention form (groups of <= 24 tokens), 26 take the four-pass form, 40 the two-key-tile kernel."""
    from oracle import d3d_oracle as orc
    from diff3dhpe_amd.synth import synth_inputs
    for J in (21, 26, 40):
        cfg = DenoiserConfig(num_frame=27, num_joints=J, embed_dim=512, depth=2)
        net, diff = build_product(cfg, 33, sampling=2, precision=prec)
        inp = {k: torch.from_numpy(v) for k, v in synth_inputs(2, 27, J, seed=330 + J).items()}
        _, y0 = diff(clean_3d_pose=torch.zeros_like(inp["noise"]).cuda(), noisy_2d_pose=inp["x2d"].cuda(), output_loss=False,
                     init_noise=inp["noise"].cuda())
        ref = orc.ddim_sample_loop(torch_sd(cfg, 33), orc.diffusion_tables("cosine", 1000), inp["x2d"], inp["noise"],
                                   num_timesteps=1000, sampling_timesteps=2, depth=cfg.depth)
        assert tuple(y0.shape) == (2, 27, J, 3) and maxabs(y0, ref) <= GATE, (J, maxabs(y0, ref))


EVAL_WORKER = """
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["D3D_ROOT"]); sys.path.insert(0, os.path.join(os.environ["D3D_ROOT"], "tests"))
from helpers import cfg_small, build_product, inputs, hashed
from diff3dhpe_amd.evaluate import evaluate
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
_, diff = build_product(cfg_small(27), 12, sampling=3, precision="f16x3")
batches = []
for B, seed in ((3, 5), (1, 6)):            # ragged over two ranks (2 + 1), then fewer windows than ranks (1 + 0)
    inp = inputs(B, 27, seed)
    mask = torch.ones(B, 27, dtype=torch.bool); mask[-1, 20:] = False
    batches.append({"inputs_2d": inp["x2d"], "inputs_3d": inp["gt3d"], "target_mask": mask, "init_noise": inp["noise"],
                    "init_noise_flip": hashed("flipnoise", tuple(inp["noise"].shape), seed)})
res = evaluate(diff, batches, scale=1.3, verbose=False)
print("RESULT " + json.dumps({"mpjpe_mm": res["mpjpe_mm"], "frames": res["frames"]}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def test_evaluate_two_ranks_with_fewer_windows_than_ranks(tmp_path):
    """evaluate() as two torch.distributed ranks sharing cuda:0 (gloo): shards 2 + 1, then 1 + 0 -- the rank without a window
    skips the sampling call (the reference's forward() cannot take an empty batch) and still joins the all-gather; every rank
    reports the one-process MPJPE and frame count, bit for bit."""
    import json, os, subprocess, sys
    from conftest import ROOT
    script = tmp_path / "eval_worker.py"
    script.write_text(EVAL_WORKER)
    env = dict(os.environ, D3D_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", OMP_NUM_THREADS="1")

    def result(out):
        return json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
    one = subprocess.run([sys.executable, str(script)], env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, WORLD_SIZE="2", RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        assert result(o) == result(one.stdout), (result(o), result(one.stdout))
    assert result(one.stdout)["frames"] == 4 * 27 - 2 * 7
