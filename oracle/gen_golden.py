#!/usr/bin/env python3
"""Golden-vector generator -- TEST INFRASTRUCTURE ONLY, runs in the BUILD container only.

Imports the *real* reference from /root/reference (CPU), drives it with the seeded synthetic
weights/inputs of ``diff3dhpe_amd.synth`` and writes input/output vectors to ``tests/golden/*.npz``.
Nothing of the reference travels: the fixtures are numbers only, and weights/inputs are re-derived
from seeds on the other side, so each file holds just the expected outputs (+ small metadata).

It also cross-checks ``oracle/d3d_oracle.py`` against the reference while it runs (max-abs printed,
hard-fails above 2e-6) -- that is how the restatement is pinned.

Usage:  python oracle/gen_golden.py [--only NAME ...]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

from diff3dhpe_amd.spec import DenoiserConfig, S2S_NAME, S2F_NAME  # noqa: E402
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs, hash_uniform  # noqa: E402
from oracle import d3d_oracle as orc  # noqa: E402


def import_reference():
    """SURVEY.md Appendix C recipe: stub timm.DropPath (never executed in eval), then import."""
    for name in ("timm", "timm.models", "timm.models.layers"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            raise RuntimeError("DropPath stub reached in training mode")

    sys.modules["timm.models.layers"].DropPath = DropPath
    sys.modules["timm"].models = sys.modules["timm.models"]
    sys.modules["timm.models"].layers = sys.modules["timm.models.layers"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from common.nets.load_net import HPE_model
    import common.conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames as d_s2s
    import common.conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames as d_s2f
    from common.loss import mpjpe
    return HPE_model, d_s2s.GaussianDiffusion, d_s2f.GaussianDiffusion, mpjpe


HPE_model, GD_S2S, GD_S2F, ref_mpjpe = import_reference()


def build_ref(cfg: DenoiserConfig, seed: int, timesteps=1000, sampling=9, eta=0.0, clip=True, family="uniform"):
    name = S2F_NAME if cfg.seq2frame else S2S_NAME
    net = HPE_model(name)(num_frame=cfg.num_frame, num_joints=cfg.num_joints, in_chans=cfg.in_chans,
                          embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                          qkv_bias=True, qk_scale=None, drop_path_rate=0.1, with_time_emb=cfg.with_time_emb)
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed, family=family).items()}
    missing, unexpected = net.load_state_dict(sd, strict=True), None
    GD = GD_S2F if cfg.seq2frame else GD_S2S
    diff = GD(model=net, timesteps=timesteps, sampling_timesteps=sampling, loss_type="l2", clip_denoised=clip,
              beta_schedule="cosine", ddim_sampling_eta=eta, clipLoss=True).eval()
    return net, diff, sd


class inject_noise:
    """Replace torch.randn / randn_like inside the unmodified reference loop (Appendix C)."""

    def __init__(self, init_noise: torch.Tensor):
        self.init = init_noise

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        init = self.init
        torch.randn = lambda *a, **k: init.clone()
        torch.randn_like = lambda t, **k: torch.zeros_like(t)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def check(tag, a: torch.Tensor, b: torch.Tensor, tol=2e-6):
    d = (a - b).abs().max().item()
    print(f"  oracle-vs-reference {tag}: max-abs {d:.3e}")
    if not d <= tol:
        raise SystemExit(f"oracle restatement diverges from the reference at {tag}: {d}")
    return d


def cfg_small(T=81, **kw):
    return DenoiserConfig(num_frame=T, embed_dim=32, depth=4, **kw)


def cfg_full(T, **kw):
    return DenoiserConfig(num_frame=T, embed_dim=512, depth=8, **kw)


# ----------------------------------------------------------------------------------------------- generators

def gen_schedules():
    net, diff, _ = build_ref(cfg_small(9), 0)
    out = {}
    for k, v in diff.state_dict().items():
        if not k.startswith("model."):
            out["cosine/" + k] = v.numpy()
    tabs = orc.diffusion_tables("cosine", 1000)
    for k in tabs:
        assert torch.equal(tabs[k], diff.state_dict()[k]), k
    for sched in ("linear", "logcosine"):
        d = GD_S2S(model=net, timesteps=1000, sampling_timesteps=9, beta_schedule=sched)
        for k in ("betas", "alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_alphas_cumprod"):
            out[f"{sched}/{k}"] = d.state_dict()[k].numpy()
            assert torch.equal(orc.diffusion_tables(sched, 1000)[k], d.state_dict()[k]), (sched, k)
    d100 = GD_S2S(model=net, timesteps=100, sampling_timesteps=20, beta_schedule="cosine")
    out["cosine100/alphas_cumprod"] = d100.state_dict()["alphas_cumprod"].numpy()
    save("schedules", **out)


def gen_ddim_times():
    flat, offs = [], [0]
    for S in range(1, 1001):
        times = torch.linspace(-1, 999, steps=S + 1)
        tl = list(reversed(times.int().tolist()))
        assert tl == orc.ddim_times(1000, S) == orc.ddim_times_scalar(1000, S), S
        flat += tl
        offs.append(len(flat))
    extra = {}
    for N, S in ((100, 20), (100, 100), (50, 7), (200, 33)):
        times = torch.linspace(-1, N - 1, steps=S + 1)
        tl = list(reversed(times.int().tolist()))
        assert tl == orc.ddim_times_scalar(N, S), (N, S)
        extra[f"N{N}_S{S}"] = np.asarray(tl, dtype=np.int16)
    save("ddim_times_N1000", flat=np.asarray(flat, dtype=np.int16), offsets=np.asarray(offs, dtype=np.int32), **extra)


def gen_temb():
    for D, depth, ts in ((32, 4, [0, 1, 110, 443, 887, 999]), (512, 8, [0, 443, 999])):
        cfg = DenoiserConfig(num_frame=9, embed_dim=D, depth=depth)
        net, _, sd = build_ref(cfg, 1)
        t = torch.tensor(ts, dtype=torch.long)
        with torch.no_grad():
            sin = net.time_mlp[0](t)
            trunk = net.time_mlp(t)
            per_block = []
            for i in range(depth):
                per_block.append(net.STEblocks[i].time_mlp(trunk))
                per_block.append(net.TTEblocks[i].time_mlp(trunk))
            per_block = torch.stack(per_block, dim=1)  # (nt, 2*depth, D) in execution order STE0,TTE0,STE1,...
        check(f"temb trunk D={D}", orc.time_trunk(sd, t, D), trunk)
        save(f"temb_D{D}", t=np.asarray(ts, np.int32), sinusoid=sin.numpy(), trunk=trunk.numpy(), per_block=per_block.numpy(),
             seed=np.int32(1))


def gen_attention():
    out = {}
    for tag, D, N, G in (("spatial_D512", 512, 17, 2), ("spatial_D32", 32, 17, 4), ("temporal_D512_T27", 512, 27, 1),
                         ("temporal_D512_T81", 512, 81, 1), ("temporal_D512_T243", 512, 243, 1),
                         ("temporal_D32_T81", 32, 81, 2)):
        cfg = DenoiserConfig(num_frame=9, embed_dim=D, depth=1)
        net, _, sd = build_ref(cfg, 2)
        x = torch.from_numpy((1.5 * hash_uniform("attn_in/" + tag, G * N * D, 2)).astype(np.float32).reshape(G, N, D))
        blk = net.STEblocks[0] if tag.startswith("spatial") else net.TTEblocks[0]
        pfx = "STEblocks.0" if tag.startswith("spatial") else "TTEblocks.0"
        with torch.no_grad():
            y = blk.attn(x)
        check("attn " + tag, orc.grand_attention(sd, pfx + ".attn", x, 8), y)
        out[tag] = y.numpy()
    save("attention", seed=np.int32(2), **out)


def gen_blocks():
    out = {}
    for tag, D, shape in (("ste_D512", 512, (1, 4, 17, 512)), ("tte_D512_T81", 512, (1, 81, 2, 512)),
                          ("ste_D32", 32, (2, 9, 17, 32)), ("tte_D32_T27", 32, (2, 27, 17, 32))):
        cfg = DenoiserConfig(num_frame=shape[1], embed_dim=D, depth=1)
        net, _, sd = build_ref(cfg, 3)
        n = int(np.prod(shape))
        x = torch.from_numpy((1.2 * hash_uniform("block_in/" + tag, n, 3)).astype(np.float32).reshape(shape))
        temb = torch.from_numpy(hash_uniform("block_temb/" + tag, shape[0] * 2 * D, 3).astype(np.float32).reshape(shape[0], 2 * D))
        spatial = tag.startswith("ste")
        blk = net.STEblocks[0] if spatial else net.TTEblocks[0]
        post = net.Spatial_norm if spatial else net.Temporal_norm
        with torch.no_grad():
            y = blk(x, is_spatial=spatial, time_emb=temb)
            z = post(y)
        yo = orc.mixste_block(sd, "STEblocks.0" if spatial else "TTEblocks.0", x, spatial, temb, 8)
        check("block " + tag, yo, y)
        out[tag + "/block"] = y.numpy()
        out[tag + "/postnorm"] = z.numpy()
    save("blocks", seed=np.int32(3), **out)


def gen_denoise():
    cases = [("small_T81", cfg_small(81), 4), ("full_T27", cfg_full(27), 2), ("full_T81", cfg_full(81), 2),
             ("full_T243", cfg_full(243), 2), ("s2f_T27", cfg_full(27, seq2frame=True), 2),
             ("notemb_T27", cfg_full(27, with_time_emb=False), 2), ("small_s2f_T27", cfg_small(27, seq2frame=True), 3)]
    for tag, cfg, B in cases:
        net, _, sd = build_ref(cfg, 4)
        inp = synth_inputs(B, cfg.num_frame, seed=100)
        x2d = torch.from_numpy(inp["x2d"])
        y_t = torch.from_numpy(inp["noise"]) * 0.7
        out = {}
        for t in (999, 443, 0):
            tv = torch.full((B,), t, dtype=torch.long)
            t0 = time.time()
            with torch.no_grad():
                r = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv)
            dt = time.time() - t0
            o = orc.forward_denoise(sd, torch.cat([x2d, y_t], dim=-1), tv, depth=cfg.depth, seq2frame=cfg.seq2frame)
            check(f"denoise {tag} t={t} ({dt:.1f}s)", o, r)
            out[f"t{t}"] = r.numpy()
        # per-row timesteps (p_losses-style call), pins the non-broadcast time-embedding path
        tv = torch.tensor([(37 * i + 5) % 1000 for i in range(B)], dtype=torch.long)
        with torch.no_grad():
            r = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv)
        out["tmixed"] = r.numpy()
        out["tmixed_t"] = tv.numpy().astype(np.int32)
        save("denoise_" + tag, seed=np.int32(4), input_seed=np.int32(100), B=np.int32(B), y_scale=np.float32(0.7), **out)


def gen_ddim():
    cases = [("small_T81_S5", cfg_small(81), 4, 5, True), ("full_T81_S9", cfg_full(81), 2, 9, False),
             ("full_T243_S9", cfg_full(243), 2, 9, False), ("full_T243_S50", cfg_full(243), 1, 50, False),
             ("s2f_T27_S9", cfg_full(27, seq2frame=True), 4, 9, True), ("full_T27_S7_notemb", cfg_full(27, with_time_emb=False), 2, 7, False),
             ("small_T81_S5_noclip", cfg_small(81), 2, 5, True)]
    for tag, cfg, B, S, traj in cases:
        clip = "noclip" not in tag
        net, diff, sd = build_ref(cfg, 5, sampling=S, clip=clip)
        inp = synth_inputs(B, cfg.num_frame, seed=200)
        x2d = torch.from_numpy(inp["x2d"])
        noise = torch.from_numpy(inp["noise"])
        if cfg.seq2frame:
            noise = noise[:, :1].contiguous()
        clean_shape = torch.zeros_like(noise)
        t0 = time.time()
        with inject_noise(noise), torch.no_grad():
            if traj:
                _, y0, rev, x0s = diff(clean_shape, x2d, None, True, False)
            else:
                _, y0 = diff(clean_3d_pose=clean_shape, noisy_2d_pose=x2d, output_loss=False)
        dt = time.time() - t0
        tabs = orc.diffusion_tables("cosine", 1000)
        o = orc.ddim_sample_loop(sd, tabs, x2d, noise, num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth,
                                 clip_denoised=clip, seq2frame=cfg.seq2frame, return_trajectory=traj)
        if traj:
            check(f"ddim {tag} y0 ({dt:.1f}s)", o[0], y0, 5e-6)
            check(f"ddim {tag} rev", o[1], rev, 5e-6)
            check(f"ddim {tag} x0s", o[2], x0s, 5e-6)
            save("ddim_" + tag, seed=np.int32(5), input_seed=np.int32(200), B=np.int32(B), S=np.int32(S), y0=y0.numpy(),
                 x_reverse_diffusion=rev.numpy(), x_start_est=x0s.numpy())
        else:
            check(f"ddim {tag} y0 ({dt:.1f}s)", o, y0, 5e-6)
            save("ddim_" + tag, seed=np.int32(5), input_seed=np.int32(200), B=np.int32(B), S=np.int32(S), y0=y0.numpy())


def gen_repeat_eta():
    """repeat_n>1 hypothesis averaging and eta>0 stochastic DDIM with supplied per-step noise (DIFF:290-297, 433-448)."""
    cfg = cfg_small(27)
    B, S, R = 2, 4, 3
    net, diff, sd = build_ref(cfg, 6, sampling=S, eta=0.5)
    inp = synth_inputs(B * R, cfg.num_frame, seed=300)
    x2d = torch.from_numpy(inp["x2d"][:B])
    noise = torch.from_numpy(inp["noise"])  # (B*R, ...)
    step_noise = [torch.from_numpy(hash_uniform(f"eta_noise/{i}", noise.numel(), 6).astype(np.float32).reshape(noise.shape)) for i in range(S)]
    it = iter(step_noise)
    _randn, _randn_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: noise.clone()
    torch.randn_like = lambda t, **k: next(it).clone()
    try:
        with torch.no_grad():
            _, y0 = diff(clean_3d_pose=torch.zeros(B, cfg.num_frame, 17, 3), noisy_2d_pose=x2d, output_loss=False, repeat_n=R)
    finally:
        torch.randn, torch.randn_like = _randn, _randn_like
    tabs = orc.diffusion_tables("cosine", 1000)
    o = orc.ddim_sample_loop(sd, tabs, x2d.repeat(R, 1, 1, 1), noise, num_timesteps=1000, sampling_timesteps=S, depth=cfg.depth,
                             eta=0.5, step_noise=step_noise)
    o = o.view(R, B, cfg.num_frame, 17, 3).mean(0)
    check("ddim repeat_n/eta", o, y0, 5e-6)
    save("ddim_small_T27_S4_eta05_rep3", seed=np.int32(6), input_seed=np.int32(300), B=np.int32(B), S=np.int32(S), R=np.int32(R),
         eta=np.float32(0.5), y0=y0.numpy())


def gen_plosses():
    cfg = cfg_small(27)
    B = 3
    net, diff, sd = build_ref(cfg, 7, sampling=5)
    inp = synth_inputs(B, cfg.num_frame, seed=400)
    x2d, noise, gt = (torch.from_numpy(inp[k]) for k in ("x2d", "noise", "gt3d"))
    gt = gt * 0.5
    t = torch.tensor([999, 12, 500], dtype=torch.long)
    _randint = torch.randint
    torch.randint = lambda *a, **k: t.clone()
    try:
        with torch.no_grad():
            loss = diff.p_losses(gt, x2d, noise=noise)
            xq = diff.q_sample(gt, t, noise)
    finally:
        torch.randint = _randint
    tabs = orc.diffusion_tables("cosine", 1000)
    lo = orc.p_losses(sd, tabs, gt, x2d, t, noise, depth=cfg.depth, clip_loss=True)
    check("p_losses", lo, loss, 5e-6)
    check("q_sample", orc.q_sample(tabs, gt, t, noise), xq)
    save("plosses_small_T27", seed=np.int32(7), input_seed=np.int32(400), B=np.int32(B), t=t.numpy().astype(np.int32),
         gt_scale=np.float32(0.5), loss=loss.numpy(), q_sample=xq.numpy())


def gen_evalmath():
    B, T, J = 3, 9, 17
    rng = np.random.RandomState(7)
    pred = torch.from_numpy(rng.uniform(-1, 1, (B, T, J, 3)).astype(np.float32))
    pred_flip = torch.from_numpy(rng.uniform(-1, 1, (B, T, J, 3)).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(-1, 1, (B, T, J, 3)).astype(np.float32))
    mask = torch.from_numpy(rng.uniform(0, 1, (B, T)) > 0.3)
    scale = 1.7
    jl, jr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    # RUN:583-590 transcribed as data flow on tensors (the harness lines themselves are script code, not importable)
    pf = pred_flip.clone()
    pf[:, :, :, 0] *= -1
    pf[:, :, jl + jr] = pf[:, :, jr + jl]
    merged = ((pred + pf) / 2.0) * scale
    merged = merged.view(-1, J, 3)[mask.view(-1) == True, :, :].unsqueeze(1)  # noqa: E712
    gtm = gt.view(-1, J, 3)[mask.view(-1) == True, :, :].unsqueeze(1)  # noqa: E712
    err = ref_mpjpe(merged, gtm)
    om = orc.merge_flip_tta(pred, pred_flip, scale, mask)
    check("evalmath merge", om, merged, 0.0)
    check("evalmath mpjpe", orc.mpjpe(om, gtm), err, 0.0)
    save("evalmath", pred=pred.numpy(), pred_flip=pred_flip.numpy(), gt=gt.numpy(), target_mask=mask.numpy(), scale=np.float32(scale),
         joints_left=np.asarray(jl, np.int32), joints_right=np.asarray(jr, np.int32), merged=merged.numpy(), mpjpe=np.float32(err.item()))


def gen_chunks():
    """Window tables straight from the reference ChunkedGenerator (out_all=True), incl. the flipped 2D copy."""
    from common.nosiy_generators import ChunkedGenerator
    out = {}
    kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    cases = [(700, 243), (243, 243), (486, 243), (487, 243), (100, 27), (81, 27), (20, 27), (1, 9)]
    for n, T in cases:
        rng = np.random.RandomState(n * 1000 + T)
        p2 = rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32)
        p3 = rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32)
        key = ("S9", "Walk", 0)
        gen = ChunkedGenerator(1, None, {key: p3}, {key: p2}, {key: np.arange(n)}, chunk_length=T, pad=0, kps_left=kl, kps_right=kr,
                               joints_left=kl, joints_right=kr, out_all=True)
        wins, flips, masks, gts = [], [], [], []
        for (seq, s3, e3, st3, et3, fl, rv) in gen.pairs:
            _, g3, w2, m, *_ = gen.get_batch_seq2seq(tuple(seq), int(s3), int(e3), int(st3), False, False)
            _, _, w2f, _, *_ = gen.get_batch_seq2seq(tuple(seq), int(s3), int(e3), int(st3), True, False)
            if m is None:
                m = np.full(T, True, dtype=bool)
            wins.append(w2); flips.append(w2f); masks.append(m); gts.append(g3)
        w, m = orc.gather_windows(torch.from_numpy(p2), T)
        wf, _ = orc.gather_windows(torch.from_numpy(p2), T, True, kl, kr)
        g, _ = orc.gather_windows(torch.from_numpy(p3), T)
        assert np.array_equal(w.numpy(), np.stack(wins)) and np.array_equal(wf.numpy(), np.stack(flips)), (n, T)
        assert np.array_equal(m.numpy(), np.stack(masks)) and np.array_equal(g.numpy(), np.stack(gts)), (n, T)
        tag = f"n{n}_T{T}"
        out[tag + "/mask"] = np.stack(masks)
        out[tag + "/starts"] = np.asarray([int(p[1]) for p in gen.pairs], np.int32)
        out[tag + "/win_checksum"] = np.float64(np.stack(wins).astype(np.float64).sum())
        out[tag + "/flip_checksum"] = np.float64((np.stack(flips).astype(np.float64) * np.arange(1, 35).reshape(17, 2)).sum())
        print(f"  chunks n={n} T={T}: {len(gen.pairs)} windows, oracle == reference")
    save("chunks", **out)


def gen_dataset():
    """Reference load_Dataset(..., 'test') + ChunkedGenerator on the synthetic H36M-shaped data set of diff3dhpe_amd.synth
    (synthetic cameras: the reference's H36M camera literals are not involved): every evaluation window in `pairs` order."""
    import tempfile
    from types import SimpleNamespace
    from common.mocap_dataset import MocapDataset
    from common.skeleton import Skeleton
    from common.camera import world_to_camera
    from data.load_noisy_data import load_Dataset
    from diff3dhpe_amd.synth import write_synth_mocap, SYNTH_PARENTS, SYNTH_JOINTS_LEFT, SYNTH_JOINTS_RIGHT
    from diff3dhpe_amd.data import EvalData, MocapMeta
    out = {}
    with tempfile.TemporaryDirectory() as root:
        positions, cameras, keypoints, meta = write_synth_mocap(root, seed=0)
        for T in (27, 9):
            ds = MocapDataset(fps=50, skeleton=Skeleton(parents=list(SYNTH_PARENTS), joints_left=list(SYNTH_JOINTS_LEFT),
                                                        joints_right=list(SYNTH_JOINTS_RIGHT)))
            ds._cameras = cameras
            ds._data = {s: {a: {"positions": p, "cameras": cameras[s]} for a, p in acts.items()} for s, acts in positions.items()}
            allp = []
            for s, acts in positions.items():          # h36m_dataset.py:260-275
                for a, p in acts.items():
                    for cam in cameras[s]:
                        allp.append(world_to_camera(p, R=cam["orientation"], t=cam["translation"]))
            allp = np.concatenate(allp, axis=0)
            cen = allp - allp[:, :1]
            ds._pos_3d_min, ds._pos_3d_max = cen.min(), cen.max()
            ds._w_mpjpe = torch.ones(17)
            opt = SimpleNamespace(dataset="h36m", keypoints="synth", subjects_train="S9", subjects_test="S9,S11", actions="*",
                                  downsample=1, subset=1, stride=T, test_time_augmentation=True, number_of_frames=T, out_all=True,
                                  batch_size=4, data_augmentation=False)
            ref = load_Dataset(opt, ds, root, "test")
            items = [ref[i] for i in range(len(ref))]
            g3 = np.stack([it[1] for it in items]); g3n = np.stack([it[2] for it in items]); x2 = np.stack([it[3] for it in items])
            x2f = np.stack([it[4] for it in items]); tm = np.stack([it[5] for it in items])
            # the product's adaptor on the same files must hand out the same windows
            ed = EvalData(MocapMeta(positions, cameras, SYNTH_JOINTS_LEFT, SYNTH_JOINTS_RIGHT), keypoints, meta["keypoints_symmetry"],
                          ["S9", "S11"], T)
            mine = list(ed.items())
            assert len(mine) == len(items), (len(mine), len(items))
            for nm, refa, key in (("3d", g3, "inputs_3d"), ("3dn", g3n, "inputs_3d_norm"), ("2d", x2, "inputs_2d"), ("2df", x2f, "inputs_2d_flip"),
                                  ("mask", tm, "target_mask")):
                m = np.stack([it[key] for it in mine])
                assert m.dtype == refa.dtype and np.array_equal(m, refa), (T, nm, m.dtype, refa.dtype, np.abs(m.astype(np.float64) - refa).max())
            assert np.float32(ed.scale) == np.float32(ref.scale), (ed.scale, ref.scale)
            if T == 27:      # whole arrays for one window length, position-weighted checksums for the other
                out[f"T{T}/inputs_3d"] = g3; out[f"T{T}/inputs_3d_norm"] = g3n; out[f"T{T}/inputs_2d"] = x2
                out[f"T{T}/inputs_2d_flip"] = x2f
            for nm, arr in (("inputs_3d", g3), ("inputs_3d_norm", g3n), ("inputs_2d", x2), ("inputs_2d_flip", x2f)):
                wts = np.arange(1, arr.size + 1, dtype=np.float64).reshape(arr.shape) % 9973.0
                out[f"T{T}/{nm}_checksum"] = np.float64((arr.astype(np.float64) * wts).sum())
            out[f"T{T}/target_mask"] = tm; out[f"T{T}/scale"] = np.float32(ref.scale)
            print(f"  dataset T={T}: {len(items)} windows, scale {float(ref.scale):.6f}: diff3dhpe_amd.data == reference (bit-equal)")
            if T == 27:
                # the runner's robustness options (--test_extra_noise_std / --test_joint_drop, RUN:730-731; LOAD:273-290), and its
                # per-action data sets (action_filter=[action_key], RUN:730): numpy's global generator seeded in front of the iteration
                noisy = {}
                for tag, kw in (("noise", dict(noise_std=0.02)), ("drop", dict(joint_drop_rate=0.15)), ("both_walk", dict(noise_std=0.05, joint_drop_rate=0.1,
                                                                                                                        action_filter=["Walk"]))):
                    refn = load_Dataset(opt, ds, root, "test", **kw)
                    np.random.seed(1234)
                    its = [refn[i] for i in range(len(refn))]
                    a2, a2f = np.stack([it[3] for it in its]), np.stack([it[4] for it in its])
                    np.random.seed(1234)
                    mine = list(ed.items(action_filter=kw.get("action_filter"), noise_std=kw.get("noise_std", 0.0),
                                         joint_drop_rate=kw.get("joint_drop_rate", 0.0)))
                    m2, m2f = np.stack([it["inputs_2d"] for it in mine]), np.stack([it["inputs_2d_flip"] for it in mine])
                    assert m2.dtype == a2.dtype and np.array_equal(m2, a2) and np.array_equal(m2f, a2f), tag
                    noisy[f"{tag}/inputs_2d"], noisy[f"{tag}/inputs_2d_flip"] = a2, a2f
                    print(f"  dataset T=27 {tag}: {len(its)} windows, diff3dhpe_amd.data == reference (bit-equal, seed 1234)")
                save("dataset_eval_noisy", seed=np.int32(1234), **noisy)
    save("dataset_eval", **out)


def gen_trainedlike():
    """Second weight family ("trained-like": heavy-tailed Linear weights, LayerNorm gains in [0.05, 6] with biases up to 3 gamma,
    position embeddings of order 1 -- diff3dhpe_amd.synth._trainedlike_param), through the imported reference: raw denoiser
    outputs at T = 27 / 243 and a full 9-step sampling at T = 81 (B = 2, D = 512, depth 8)."""
    for T, ts in ((27, (999, 443, 0)), (243, (443,))):
        cfg = cfg_full(T)
        net, _, sd = build_ref(cfg, 11, family="trainedlike")
        inp = synth_inputs(2, T, seed=500)
        x2d = torch.from_numpy(inp["x2d"])
        y_t = torch.from_numpy(inp["noise"]) * 0.7
        out = {}
        for t in ts:
            tv = torch.full((2,), t, dtype=torch.long)
            t0 = time.time()
            with torch.no_grad():
                r = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv)
            dt = time.time() - t0
            o = orc.forward_denoise(sd, torch.cat([x2d, y_t], dim=-1), tv, depth=cfg.depth)
            check(f"denoise trainedlike T={T} t={t} ({dt:.1f}s)", o, r)
            out[f"t{t}"] = r.numpy()
        if T == 27:
            tv = torch.tensor([905, 17], dtype=torch.long)
            with torch.no_grad():
                out["tmixed"] = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv).numpy()
            out["tmixed_t"] = tv.numpy().astype(np.int32)
        save(f"denoise_trainedlike_T{T}", seed=np.int32(11), input_seed=np.int32(500), B=np.int32(2), y_scale=np.float32(0.7), **out)
    cfg = cfg_full(81)
    net, diff, sd = build_ref(cfg, 12, sampling=9, family="trainedlike")
    inp = synth_inputs(2, 81, seed=600)
    x2d, noise = torch.from_numpy(inp["x2d"]), torch.from_numpy(inp["noise"])
    t0 = time.time()
    with inject_noise(noise), torch.no_grad():
        _, y0, rev, x0s = diff(torch.zeros_like(noise), x2d, None, True, False)
    dt = time.time() - t0
    o = orc.ddim_sample_loop(sd, orc.diffusion_tables("cosine", 1000), x2d, noise, num_timesteps=1000, sampling_timesteps=9, depth=8,
                             return_trajectory=True)
    check(f"ddim trainedlike T=81 S=9 y0 ({dt:.1f}s)", o[0], y0, 5e-6)
    check("ddim trainedlike x0s", o[2], x0s, 5e-6)
    save("ddim_trainedlike_T81_S9", seed=np.int32(12), input_seed=np.int32(600), B=np.int32(2), S=np.int32(9), y0=y0.numpy(),
         x_start_est=x0s.numpy())


def gen_round4():
    """The two reachable constructor combinations that had no fixture: seq2frame WITHOUT time embedding (the
    reference's own 3DHP command lines: Experiments.sh:15-17, --model ...S2F... with_time_emb False) as raw denoiser outputs and
    as a full 7-step sampling, and the trained-like weight family on the seq2frame model.  All through the imported reference."""
    def denoise_case(tag, cfg, seed, family, in_seed):
        net, _, sd = build_ref(cfg, seed, family=family)
        inp = synth_inputs(2, cfg.num_frame, seed=in_seed)
        x2d = torch.from_numpy(inp["x2d"])
        y_t = torch.from_numpy(inp["noise"]) * 0.7
        out = {}
        for t in (999, 443, 0):
            tv = torch.full((2,), t, dtype=torch.long)
            with torch.no_grad():
                r = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv)
            o = orc.forward_denoise(sd, torch.cat([x2d, y_t], dim=-1), tv, depth=cfg.depth, seq2frame=cfg.seq2frame)
            check(f"denoise {tag} t={t}", o, r)
            out[f"t{t}"] = r.numpy()
        tv = torch.tensor([905, 17], dtype=torch.long)
        with torch.no_grad():
            out["tmixed"] = net.forward_denoise(torch.cat([x2d, y_t], dim=-1), tv).numpy()
        out["tmixed_t"] = tv.numpy().astype(np.int32)
        save("denoise_" + tag, seed=np.int32(seed), input_seed=np.int32(in_seed), B=np.int32(2), y_scale=np.float32(0.7), **out)

    denoise_case("s2f_notemb_T27", cfg_full(27, seq2frame=True, with_time_emb=False), 4, "uniform", 100)
    denoise_case("trainedlike_s2f_T27", cfg_full(27, seq2frame=True), 11, "trainedlike", 500)
    # full sampling on the 3DHP configuration: seq2frame, no time embedding, 7 DDIM steps (Experiments.sh:17)
    cfg = cfg_full(27, seq2frame=True, with_time_emb=False)
    net, diff, sd = build_ref(cfg, 6, sampling=7)
    inp = synth_inputs(3, 27, seed=700)
    x2d, noise = torch.from_numpy(inp["x2d"]), torch.from_numpy(inp["noise"][:, :1].copy())
    with inject_noise(noise), torch.no_grad():
        _, y0 = diff(torch.zeros_like(noise), x2d, None, False, False)
    o = orc.ddim_sample_loop(sd, orc.diffusion_tables("cosine", 1000), x2d, noise, num_timesteps=1000, sampling_timesteps=7, depth=8,
                             seq2frame=True)
    check("ddim s2f notemb T=27 S=7 y0", o, y0, 5e-6)
    save("ddim_s2f_notemb_T27_S7", seed=np.int32(6), input_seed=np.int32(700), B=np.int32(3), S=np.int32(7), y0=y0.numpy())


def gen_round6():
    """The three constructor arguments the engine used to refuse (S2S:140-142, 184): qkv_bias=False, a qk_scale
    override, and a norm_layer with another eps -- raw denoiser outputs of the imported reference at T = 27, full width, depth 2."""
    from functools import partial
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=2)
    inp = synth_inputs(2, 27, seed=900)
    xcat = torch.cat([torch.from_numpy(inp["x2d"]), torch.from_numpy(inp["noise"]) * 0.7], dim=-1)
    out = {}
    for tag, kw, okw in (("nobias", dict(qkv_bias=False), {}), ("qkscale", dict(qk_scale=0.2), dict(qk_scale=0.2)),
                         ("eps", dict(norm_layer=partial(nn.LayerNorm, eps=1e-3)), dict(norm_eps=1e-3))):
        args = dict(num_frame=27, num_joints=17, in_chans=2, embed_dim=512, depth=2, num_heads=8, mlp_ratio=2., qkv_bias=True, qk_scale=None,
                    drop_path_rate=0.1, with_time_emb=True)
        args.update(kw)
        net = HPE_model(S2S_NAME)(**args).eval()
        sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 13, family="trainedlike" if tag == "eps" else "uniform").items()}
        if tag == "nobias":
            sd = {k: v for k, v in sd.items() if not k.endswith("attn.qkv.bias")}
        net.load_state_dict(sd, strict=True)
        for t in (999, 17):
            tv = torch.full((2,), t, dtype=torch.long)
            with torch.no_grad():
                r = net.forward_denoise(xcat, tv)
            check(f"ctor {tag} t={t}", orc.forward_denoise(sd, xcat, tv, depth=2, **okw), r)
            out[f"{tag}_t{t}"] = r.numpy()
    save("denoise_ctor_args_T27", seed=np.int32(13), input_seed=np.int32(900), B=np.int32(2), y_scale=np.float32(0.7), qk_scale=np.float32(0.2),
         norm_eps=np.float32(1e-3), **out)


def gen_round5():
    """BASELINE configs[4] (MPI-INF-3DHP, T=27, seq2frame) end to end, from the reference itself:
      chunks_s2f        window tables of ChunkedGenerator_3dhp (out_all False, stride 1; out_all True with `valid` flags)
      dataset_3dhp_eval MPIINF3DHPDataset + load_Dataset_3dhp(split='test') on the synthetic 3DHP-shaped files of diff3dhpe_amd.synth:
                        every evaluation item in `pairs` order, both window tables; diff3dhpe_amd.data.EvalData3DHP must be bit-equal
      evaluate_3dhp_s2f the data flow of the 3DHP runner's evaluate() (run_..._3dhp.py:510-533: output_loss=True default, flipped clean
                        pose, un-flip / average / de-normalise / mask / mpjpe) on two DataLoader batches of that data set with the
                        reference S2F model WITHOUT time embedding (Experiments.sh:15-17) -- per-batch MPJPE and frame counts"""
    import tempfile
    from types import SimpleNamespace
    from common.nosiy_generators import ChunkedGenerator_3dhp
    from common.mpiinf3dhp_dataset import MPIINF3DHPDataset
    from data.load_noisy_data import load_Dataset_3dhp
    from diff3dhpe_amd.synth import write_synth_3dhp
    from diff3dhpe_amd.data import EvalData3DHP

    # ---- window tables
    out = {}
    kl, kr = [5, 6, 7, 11, 12, 13], [2, 3, 4, 8, 9, 10]
    for n, T in [(100, 27), (27, 27), (5, 27), (1, 9), (40, 9)]:
        rng = np.random.RandomState(n * 977 + T)
        p2 = rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32)
        p3 = rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32)
        valid = (rng.uniform(0, 1, n) > 0.3).astype(np.float64)
        gen = ChunkedGenerator_3dhp(1, None, {"TS1": p3}, {"TS1": p2}, 1, pad=(T - 1) // 2, kps_left=kl, kps_right=kr, joints_left=kl,
                                    joints_right=kr, out_all=False, valid_frame={"TS1": valid}, split="test")
        wins, flips, gts, masks = [], [], [], []
        for (seq, s3, e3, fl, rv) in gen.pairs:
            _, g3, w2, m, *_ = gen.get_batch_seq2frame(seq, int(s3), int(e3), False, False)
            _, _, w2f, _, *_ = gen.get_batch_seq2frame(seq, int(s3), int(e3), True, False)
            wins.append(w2); flips.append(w2f); gts.append(g3); masks.append(m)
        w, g, m = orc.gather_windows_s2f(torch.from_numpy(p2), torch.from_numpy(p3), valid, T)
        wf, _, _ = orc.gather_windows_s2f(torch.from_numpy(p2), torch.from_numpy(p3), valid, T, True, kl, kr)
        assert np.array_equal(w.numpy(), np.stack(wins)) and np.array_equal(wf.numpy(), np.stack(flips)), (n, T)
        assert np.array_equal(g.numpy(), np.stack(gts)) and np.array_equal(m.numpy(), np.stack(masks)), (n, T)
        tag = f"s2f_n{n}_T{T}"
        out[tag + "/mask"] = np.stack(masks)
        out[tag + "/win_checksum"] = np.float64((np.stack(wins).astype(np.float64) * (np.arange(1, T * 34 + 1).reshape(T, 17, 2) % 97)).sum())
        out[tag + "/flip_checksum"] = np.float64((np.stack(flips).astype(np.float64) * (np.arange(1, T * 34 + 1).reshape(T, 17, 2) % 89)).sum())
        print(f"  chunks s2f n={n} T={T}: {len(gen.pairs)} windows, oracle == reference")
    save("chunks_s2f", **out)

    # ---- the data set through the reference's loaders
    out, ev = {}, {}
    with tempfile.TemporaryDirectory() as root:
        test, train = write_synth_3dhp(root, seed=0)
        for oa, T in ((False, 27), (True, 27), (False, 9)):
            opt = SimpleNamespace(dataset="3dhp", keypoints="gt", subjects_train="S1,S2", subjects_test="TS1,TS5", actions="*", downsample=1,
                                  subset=1, stride=(T if oa else 1), test_time_augmentation=True, number_of_frames=T, out_all=oa,
                                  batch_size=4, data_augmentation=False)
            ds = MPIINF3DHPDataset(opt, root_path=root)
            ref = load_Dataset_3dhp(opt, ds._test, pos_3d_min=ds._pos_3d_min, pos_3d_max=ds._pos_3d_max, split="test")
            items = [ref[i] for i in range(len(ref))]
            g3 = np.stack([it[1] for it in items]); g3n = np.stack([it[2] for it in items]); x2 = np.stack([it[3] for it in items])
            x2f = np.stack([it[4] for it in items]); tm = np.stack([it[5] for it in items])
            ed = EvalData3DHP(test, ["TS1", "TS5"], T, out_all=oa, train_data=train)
            mine = list(ed.items())
            assert len(mine) == len(items) == len(ed), (len(mine), len(items))
            for nm, refa, key in (("3d", g3, "inputs_3d"), ("3dn", g3n, "inputs_3d_norm"), ("2d", x2, "inputs_2d"), ("2df", x2f, "inputs_2d_flip"),
                                  ("mask", tm, "target_mask")):
                m = np.stack([it[key] for it in mine])
                assert m.dtype == refa.dtype and m.shape == refa.shape and np.array_equal(m, refa), (oa, T, nm, m.dtype, refa.dtype, m.shape, refa.shape)
            assert np.float32(ed.scale) == np.float32(ref.scale), (ed.scale, ref.scale)
            tag = f"{'s2s' if oa else 's2f'}_T{T}"
            if T == 27:
                out[tag + "/inputs_3d"] = g3; out[tag + "/inputs_2d"] = x2
            for nm, arr in (("inputs_3d", g3), ("inputs_3d_norm", g3n), ("inputs_2d", x2), ("inputs_2d_flip", x2f)):
                wts = np.arange(1, arr.size + 1, dtype=np.float64).reshape(arr.shape) % 9973.0
                out[f"{tag}/{nm}_checksum"] = np.float64((arr.astype(np.float64) * wts).sum())
            out[tag + "/target_mask"] = tm; out[tag + "/scale"] = np.float32(ref.scale)
            print(f"  3dhp {tag}: {len(items)} items, scale {float(ref.scale):.3f}: diff3dhpe_amd.data.EvalData3DHP == reference (bit-equal)")

            if (oa, T) != (False, 27):
                continue
            # ---- evaluate(): the 3DHP runner's data flow on DataLoader batches of this data set, seq_filter='TS1' as run_evaluation() does
            cfg = cfg_full(27, seq2frame=True, with_time_emb=False)
            S = 3
            net, diff, sd = build_ref(cfg, 11, sampling=S)
            refq = load_Dataset_3dhp(opt, ds._test, pos_3d_min=ds._pos_3d_min, pos_3d_max=ds._pos_3d_max, split="test", seq_filter="TS1")
            loader = torch.utils.data.DataLoader(refq, batch_size=32, shuffle=False, num_workers=0, drop_last=False)
            jl, jr = ds.joints_left, ds.joints_right
            tabs = orc.diffusion_tables("cosine", 1000)
            errs, cnts = [], []
            for bi, (_, inputs_3d, inputs_3d_norm, inputs_2d, inputs_2d_flip, target_mask, *_rest) in enumerate(loader):
                B = inputs_2d.shape[0]
                noise = torch.from_numpy(hash_uniform(f"eval3dhp/noise/{bi}", B * 17 * 3, 5).astype(np.float32).reshape(B, 1, 17, 3)) * 1.7
                noise_f = torch.from_numpy(hash_uniform(f"eval3dhp/noise_flip/{bi}", B * 17 * 3, 5).astype(np.float32).reshape(B, 1, 17, 3)) * 1.7
                tmask = target_mask.view(-1)
                n3f = inputs_3d_norm.clone()
                n3f[:, :, :, 0] *= -1
                n3f[:, :, jl + jr] = n3f[:, :, jr + jl]
                with torch.no_grad():
                    with inject_noise(noise):
                        _, pred = diff(clean_3d_pose=inputs_3d_norm, noisy_2d_pose=inputs_2d)              # output_loss=True: the default
                    with inject_noise(noise_f):
                        _, pred_f = diff(clean_3d_pose=n3f, noisy_2d_pose=inputs_2d_flip)
                pred_f[:, :, :, 0] *= -1
                pred_f[:, :, jl + jr] = pred_f[:, :, jr + jl]
                p = (pred + pred_f) / 2.0
                p = refq.reverse_norm_3d_pose(p)
                p = p.view(-1, 17, 3)[tmask == True, :, :].unsqueeze(1)  # noqa: E712
                g = inputs_3d.view(-1, 17, 3)[tmask == True, :, :].unsqueeze(1)  # noqa: E712
                e = ref_mpjpe(p, g)
                oe, on = orc.evaluate_batch(sd, tabs, inputs_2d, inputs_2d_flip, inputs_3d, target_mask, noise, noise_f, scale=float(refq.scale),
                                            depth=8, sampling_timesteps=S, seq2frame=True, joints_left=jl, joints_right=jr)
                assert on == g.shape[0]
                check(f"3dhp evaluate batch {bi} mpjpe (mm)", oe, e, 2e-3)
                errs.append(float(e)); cnts.append(int(g.shape[0]))
            ev = dict(seed=np.int32(11), S=np.int32(S), batch_size=np.int32(32), mpjpe_per_batch=np.asarray(errs, np.float64),
                      frames_per_batch=np.asarray(cnts, np.int32), scale=np.float32(refq.scale))
            print(f"  3dhp evaluate(): {len(errs)} batches, {sum(cnts)} valid frames, MPJPE {np.dot(errs, cnts) / sum(cnts):.4f} mm")
    save("dataset_3dhp_eval", **out)
    save("evaluate_3dhp_s2f", **ev)


def gen_round5b():
    """The reference's OWN 3DHP command lines (Experiments.sh:15-17: ...S2S... model, out_all=True, stride = frames = 27, no time
    embedding) through the data flow of the 3DHP runner's evaluate() (run_..._3dhp.py:510-533) on the synthetic 3DHP-shaped set: seq2seq
    windows with the last one shifted, the overlap mask ANDed with the frames' `valid` flags -- per-batch MPJPE and frame counts."""
    import tempfile
    from types import SimpleNamespace
    from common.mpiinf3dhp_dataset import MPIINF3DHPDataset
    from data.load_noisy_data import load_Dataset_3dhp
    from diff3dhpe_amd.synth import write_synth_3dhp
    with tempfile.TemporaryDirectory() as root:
        write_synth_3dhp(root, seed=0)
        T, S = 27, 3
        opt = SimpleNamespace(dataset="3dhp", keypoints="gt", subjects_train="S1,S2", subjects_test="TS1,TS5,TS3", actions="*", downsample=1,
                              subset=1, stride=T, test_time_augmentation=True, number_of_frames=T, out_all=True, batch_size=4,
                              data_augmentation=False)
        ds = MPIINF3DHPDataset(opt, root_path=root)
        cfg = cfg_full(T, with_time_emb=False)
        net, diff, sd = build_ref(cfg, 12, sampling=S)
        jl, jr = ds.joints_left, ds.joints_right
        tabs = orc.diffusion_tables("cosine", 1000)
        out = {}
        for seq in ("TS1", "TS5", "TS3"):
            refq = load_Dataset_3dhp(opt, ds._test, pos_3d_min=ds._pos_3d_min, pos_3d_max=ds._pos_3d_max, split="test", seq_filter=seq)
            loader = torch.utils.data.DataLoader(refq, batch_size=2, shuffle=False, num_workers=0, drop_last=False)
            errs, cnts = [], []
            for bi, (_, inputs_3d, inputs_3d_norm, inputs_2d, inputs_2d_flip, target_mask, *_rest) in enumerate(loader):
                B = inputs_2d.shape[0]
                noise = torch.from_numpy(hash_uniform(f"eval3dhp_s2s/{seq}/noise/{bi}", B * T * 17 * 3, 5).astype(np.float32).reshape(B, T, 17, 3)) * 1.7
                noise_f = torch.from_numpy(hash_uniform(f"eval3dhp_s2s/{seq}/noise_flip/{bi}", B * T * 17 * 3, 5).astype(np.float32).reshape(B, T, 17, 3)) * 1.7
                tmask = target_mask.view(-1)
                n3f = inputs_3d_norm.clone()
                n3f[:, :, :, 0] *= -1
                n3f[:, :, jl + jr] = n3f[:, :, jr + jl]
                with torch.no_grad():
                    with inject_noise(noise):
                        _, pred = diff(clean_3d_pose=inputs_3d_norm, noisy_2d_pose=inputs_2d)
                    with inject_noise(noise_f):
                        _, pred_f = diff(clean_3d_pose=n3f, noisy_2d_pose=inputs_2d_flip)
                pred_f[:, :, :, 0] *= -1
                pred_f[:, :, jl + jr] = pred_f[:, :, jr + jl]
                p = refq.reverse_norm_3d_pose((pred + pred_f) / 2.0)
                p = p.view(-1, 17, 3)[tmask == True, :, :].unsqueeze(1)  # noqa: E712
                g = inputs_3d.view(-1, 17, 3)[tmask == True, :, :].unsqueeze(1)  # noqa: E712
                e = ref_mpjpe(p, g)
                oe, on = orc.evaluate_batch(sd, tabs, inputs_2d, inputs_2d_flip, inputs_3d, target_mask, noise, noise_f, scale=float(refq.scale),
                                            depth=8, sampling_timesteps=S, joints_left=jl, joints_right=jr)
                assert on == g.shape[0]
                check(f"3dhp s2s evaluate {seq} batch {bi} mpjpe (mm)", oe, e, 2e-3)
                errs.append(float(e)); cnts.append(int(g.shape[0]))
            out[f"{seq}/mpjpe_per_batch"] = np.asarray(errs, np.float64)
            out[f"{seq}/frames_per_batch"] = np.asarray(cnts, np.int32)
            print(f"  3dhp s2s evaluate() {seq}: {len(errs)} batches, {sum(cnts)} valid frames, MPJPE {np.dot(errs, cnts) / sum(cnts):.4f} mm")
        save("evaluate_3dhp_s2s", seed=np.int32(12), S=np.int32(S), batch_size=np.int32(2), scale=np.float32(refq.scale), **out)


def gen_3dhp_noisy():
    """The 3DHP runner's robustness options (--test_extra_noise_std / --test_joint_drop, run_..._3dhp.py:598-600; LOAD:422-440) through the
    reference's load_Dataset_3dhp on the synthetic 3DHP-shaped files, numpy's global generator seeded in front of the iteration: both
    window tables, with a per-sequence data set (seq_filter) for one of them."""
    import tempfile
    from types import SimpleNamespace
    from common.mpiinf3dhp_dataset import MPIINF3DHPDataset
    from data.load_noisy_data import load_Dataset_3dhp
    from diff3dhpe_amd.synth import write_synth_3dhp
    from diff3dhpe_amd.data import EvalData3DHP
    out = {}
    with tempfile.TemporaryDirectory() as root:
        test, train = write_synth_3dhp(root, seed=0)
        for oa, T, kw in ((False, 27, dict(noise_std=0.03, joint_drop_rate=0.1)), (True, 27, dict(noise_std=0.02)),
                          (False, 9, dict(joint_drop_rate=0.2, seq_filter="TS5"))):
            opt = SimpleNamespace(dataset="3dhp", keypoints="gt", subjects_train="S1,S2", subjects_test="TS1,TS5", actions="*", downsample=1,
                                  subset=1, stride=(T if oa else 1), test_time_augmentation=True, number_of_frames=T, out_all=oa,
                                  batch_size=4, data_augmentation=False)
            ds = MPIINF3DHPDataset(opt, root_path=root)
            ref = load_Dataset_3dhp(opt, ds._test, pos_3d_min=ds._pos_3d_min, pos_3d_max=ds._pos_3d_max, split="test", **kw)
            np.random.seed(4321)
            its = [ref[i] for i in range(len(ref))]
            a2, a2f = np.stack([it[3] for it in its]), np.stack([it[4] for it in its])
            ed = EvalData3DHP(test, ["TS1", "TS5"], T, out_all=oa, train_data=train)
            np.random.seed(4321)
            mine = list(ed.items(seq_filter=kw.get("seq_filter"), noise_std=kw.get("noise_std", 0.0), joint_drop_rate=kw.get("joint_drop_rate", 0.0)))
            m2, m2f = np.stack([it["inputs_2d"] for it in mine]), np.stack([it["inputs_2d_flip"] for it in mine])
            tag = f"{'s2s' if oa else 's2f'}_T{T}"
            assert m2.dtype == a2.dtype and np.array_equal(m2, a2) and np.array_equal(m2f, a2f), tag
            wts = np.arange(1, a2.size + 1, dtype=np.float64).reshape(a2.shape) % 9973.0
            out[tag + "/inputs_2d_checksum"] = np.float64((a2.astype(np.float64) * wts).sum())
            out[tag + "/inputs_2d_flip_checksum"] = np.float64((a2f.astype(np.float64) * wts).sum())
            out[tag + "/first"] = a2[:3]
            out[tag + "/zeros"] = np.int64((a2 == 0).sum())
            print(f"  3dhp noisy {tag} {kw}: {len(its)} items, diff3dhpe_amd.data.EvalData3DHP == reference (bit-equal, seed 4321)")
    save("dataset_3dhp_eval_noisy", seed=np.int32(4321), **out)


def gen_metrics():
    """evaluate()'s other three protocols (RUN:602-614): the REAL common.loss.p_mpjpe / n_mpjpe / mean_velocity_error on merged, masked
    batches as evaluate() hands them over -- (N, 1, J, 3) torch tensors for n_mpjpe, (N, J, 3) float32 numpy arrays for the other two.
    Families: a prediction near its target, an unrelated one, mirrored predictions (the det R < 0 branch of p_mpjpe), planar poses
    (a singular H), one kept frame (mean_velocity_error of an empty difference: nan), a T = 243 sized batch."""
    import warnings
    from common.loss import p_mpjpe as ref_p, n_mpjpe as ref_n, mean_velocity_error as ref_v
    rng = np.random.RandomState(11)
    J = 17
    cases = {}
    gt = rng.uniform(-1, 1, (40, J, 3)).astype(np.float32)
    cases["near"] = (gt + 0.05 * rng.standard_normal(gt.shape).astype(np.float32), gt)
    cases["unrelated"] = (rng.uniform(-1, 1, gt.shape).astype(np.float32), gt)
    mir = gt.copy()
    mir[..., 0] *= -1
    cases["mirrored"] = (mir + 0.02 * rng.standard_normal(gt.shape).astype(np.float32), gt)
    flat = gt.copy()
    flat[..., 2] = 0.25
    cases["planar"] = (flat + np.array([0.0, 0.0, 0.0], np.float32), (flat * 1.3 + 0.1).astype(np.float32))
    cases["planar"] = (cases["planar"][0] + 0.03 * rng.standard_normal(gt.shape).astype(np.float32) * np.array([1, 1, 0], np.float32), cases["planar"][1])
    cases["one_frame"] = (cases["near"][0][:1].copy(), gt[:1].copy())
    big = (1000.0 * np.cumsum(0.01 * rng.standard_normal((729, J, 3)), axis=0)).astype(np.float32)
    cases["big_mm"] = (big + (30.0 * rng.standard_normal(big.shape)).astype(np.float32), big)
    out = {}
    for tag, (pr, tg) in cases.items():
        pt, tt = torch.from_numpy(pr).unsqueeze(1), torch.from_numpy(tg).unsqueeze(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e1 = ref_mpjpe(pt, tt).item()
            e3 = ref_n(pt, tt).item()
            e2 = float(ref_p(pr.copy(), tg.copy()))
            ev = float(ref_v(pr.copy(), tg.copy()))
        n, s1, s2, s3, sv = orc.protocol_sums(pt, tt)
        for name, a, b in (("mpjpe", s1 / n, e1), ("p_mpjpe", s2 / n, e2), ("n_mpjpe", s3 / n, e3), ("mpjve", sv / n, ev)):
            same = (a == b) or (np.isnan(a) and np.isnan(b))
            print(f"  metrics {tag} {name}: reference {b:.9g} oracle {a:.9g} {'==' if same else 'DIFFERENT'}")
            if not same:
                raise SystemExit(f"oracle restatement of {name} diverges from the reference on '{tag}'")
        out[f"{tag}_pred"], out[f"{tag}_gt"] = pr, tg
        out[f"{tag}_ref"] = np.asarray([e1, e2, e3, ev], np.float64)
    save("pose_metrics", tags=np.asarray(list(cases)), **out)


GENERATORS = {
    "schedules": gen_schedules, "ddim_times": gen_ddim_times, "temb": gen_temb, "attention": gen_attention, "blocks": gen_blocks,
    "denoise": gen_denoise, "ddim": gen_ddim, "repeat_eta": gen_repeat_eta, "plosses": gen_plosses, "evalmath": gen_evalmath, "chunks": gen_chunks,
    "dataset": gen_dataset, "trainedlike": gen_trainedlike, "round4": gen_round4, "round5": gen_round5, "round5b": gen_round5b, "round6": gen_round6, "metrics": gen_metrics, "3dhp_noisy": gen_3dhp_noisy,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    for name, fn in GENERATORS.items():
        if a.only and name not in a.only:
            continue
        print(f"[{name}]")
        fn()
