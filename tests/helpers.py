"""Shared test helpers: seeded weights/inputs (never the reference; never shipped vectors of weights)."""
import numpy as np
import torch

from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict, synth_inputs, hash_uniform


def cfg_small(T=81, **kw):
    return DenoiserConfig(num_frame=T, embed_dim=32, depth=4, **kw)


def cfg_full(T, **kw):
    return DenoiserConfig(num_frame=T, embed_dim=512, depth=8, **kw)


def torch_sd(cfg, seed):
    return {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed).items()}


def inputs(B, T, seed):
    d = synth_inputs(B, T, seed=seed)
    return {k: torch.from_numpy(v) for k, v in d.items()}


def hashed(name, shape, seed, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((scale * hash_uniform(name, n, seed)).astype(np.float32).reshape(shape))


def build_product(cfg, seed, sampling=9, eta=0.0, clip=True, device="cuda", precision="fp32"):
    """HPE_model + GaussianDiffusion of the product with the seeded weights, on `device`."""
    import diff3dhpe_amd as d3d
    name = d3d.S2F_NAME if cfg.seq2frame else d3d.S2S_NAME
    net = d3d.HPE_model(name)(num_frame=cfg.num_frame, num_joints=cfg.num_joints, in_chans=cfg.in_chans,
                              embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                              qkv_bias=True, qk_scale=None, drop_path_rate=0.1, with_time_emb=cfg.with_time_emb)
    net.load_state_dict(torch_sd(cfg, seed), strict=True)
    net.precision = precision
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=sampling, loss_type="l2",
                                 clip_denoised=clip, beta_schedule="cosine", ddim_sampling_eta=eta, clipLoss=True).eval()
    if device != "cpu":
        diff = diff.to(device)
    return net, diff


def maxabs(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()
