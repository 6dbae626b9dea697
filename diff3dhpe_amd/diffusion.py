"""Host-side mirror of the reference's ``GaussianDiffusion`` (seq2seq and seq2frame variants).

Reference: common/conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames.py (DIFF) and
common/conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames.py (DIFF-S2F).
Constructor signature, registered buffers, ``forward`` calling convention (keyword and positional) and return tuples
are the reference's; the sampling loop itself (DIFF:262-300) is ONE call into libd3d_hip (`d3d_ddim_sample`).
The only host arithmetic left here is the init-time fp64 schedule tables (DIFF:114-183).
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .engine import hypothesis_mean, repeat_batch
from .nets import _MixSTEDenoiser


def _betas(name: str, timesteps: int) -> torch.Tensor:
    """fp64 beta tables: linear (DIFF:52-55), cosine (DIFF:58-68), logcosine (DIFF:70-81)."""
    if name == "linear":
        return torch.linspace(1e-4, 2e-2, timesteps, dtype=torch.float64)
    if name == "cosine":
        grid = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    elif name == "logcosine":
        grid = torch.logspace(0, 2, timesteps + 1, dtype=torch.float64) / 1e-1 / timesteps
    else:
        raise ValueError(f"unknown beta schedule {name}")
    s = 0.008
    abar = torch.cos((grid + s) / (1 + s) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)


class GaussianDiffusion(nn.Module):
    def __init__(self, model, timesteps=100, sampling_timesteps=20, loss_type='l1', conditional=True,
                 clip_denoised=False, beta_schedule='cosine', p2_loss_weight_gamma=0., p2_loss_weight_k=1,
                 ddim_sampling_eta=0., clipLoss=False):
        super().__init__()
        if not isinstance(model, _MixSTEDenoiser):
            raise TypeError("model must come from diff3dhpe_amd.HPE_model(...): the DDIM loop runs inside the HIP engine")
        if not conditional:
            raise NotImplementedError("conditional=False is not supported (every reference entry point is conditional)")
        self.model = model
        self.conditional = conditional
        self.clip_denoised = clip_denoised
        self.clipLoss = clipLoss
        self.seq2frame = bool(model.cfg.seq2frame)

        betas = _betas(beta_schedule, timesteps)
        alphas = 1. - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.)
        self.sqrt_alphas_cumprod_prev = torch.sqrt(F.pad(ac, (1, 0), value=1.))  # fp64 plain attribute (DIFF:133)

        self.num_timesteps = int(betas.shape[0])
        self.loss_type = loss_type
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = ddim_sampling_eta

        post_var = betas * (1. - ac_prev) / (1. - ac)
        tables = {
            'betas': betas,
            'alphas_cumprod': ac,
            'alphas_cumprod_prev': ac_prev,
            'sqrt_recip_alphas': torch.sqrt(1.0 / alphas),
            'sqrt_alphas_cumprod': torch.sqrt(ac),
            'sqrt_one_minus_alphas_cumprod': torch.sqrt(1. - ac),
            'log_one_minus_alphas_cumprod': torch.log(1. - ac),
            'sqrt_recip_alphas_cumprod': torch.sqrt(1. / ac),
            'sqrt_recipm1_alphas_cumprod': torch.sqrt(1. / ac - 1),
            'posterior_variance': post_var,
            'posterior_log_variance_clipped': torch.log(post_var.clamp(min=1e-20)),
            'posterior_mean_coef1': betas * torch.sqrt(ac_prev) / (1. - ac),
            'posterior_mean_coef2': (1. - ac_prev) * torch.sqrt(alphas) / (1. - ac),
            'p2_loss_weight': (p2_loss_weight_k + ac / (1 - ac)) ** -p2_loss_weight_gamma,
        }
        for name, val in tables.items():  # registered as fp32 in this order (DIFF:149-183)
            self.register_buffer(name, val.to(torch.float32))
        self._sched_sig = {}
        # The F16X3 range guard is READ after every engine call of this class (sampling, forward_denoise, p_losses) unless the model's
        # range_check is off (D3D_CHECK_RANGE=0): nets._MixSTEDenoiser._guarded -- precision "auto" repeats a flagged call on the
        # exact-fp32 engine, "f16x3" raises D3DError.

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self, device: torch.device, fallback: bool = False):
        eng = self.model.engine_for(device, fallback)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        sig = (id(eng), (self.model._engine_sig_fb if fallback else self.model._engine_sig).get(idx),
               self.sampling_timesteps, float(self.ddim_sampling_eta), bool(self.clip_denoised),
               self.alphas_cumprod.data_ptr(), self.alphas_cumprod._version)
        key = id(eng)
        if self._sched_sig.get(key) != sig or eng.sampling_timesteps is None:
            eng.set_schedule(self.alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.sampling_timesteps,
                             self.ddim_sampling_eta, self.clip_denoised, sqrt_alphas_cumprod=self.sqrt_alphas_cumprod)
            self._sched_sig[key] = sig
        return eng

    def ddim_times(self) -> List[int]:
        """Reversed integer schedule (DIFF:270-272), from the library's bit-exact host routine."""
        return _lib.ddim_times(self.num_timesteps, self.sampling_timesteps)

    # ------------------------------------------------------------------ sampling (DIFF:250-355)
    @torch.no_grad()
    def ddim_sample(self, x, t, condition_x=None):
        """One denoiser evaluation + optional clamp (DIFF:250-258). x: noisy 3D pose, t: python int."""
        time = torch.full((x.shape[0],), t, device=x.device, dtype=torch.long)
        x_start = self.model.forward_denoise(torch.cat([condition_x, x], dim=-1), time)
        return torch.clamp(x_start, min=-1., max=1.) if self.clip_denoised else x_start

    def _draw(self, x_in, target_shape, init_noise, step_noise):
        """Noise for one sampling.  When the caller supplies none, the global generator is consumed exactly as the reference
        consumes it -- one torch.randn(target_shape) (DIFF:275), then one randn_like per non-final step, S - 1 draws, whether
        or not eta multiplies them by zero (DIFF:293) -- so that under the unchanged runner (torch.manual_seed, RUN) the
        second sampling of a batch (flip-TTA), later batches and repeat_n start from the same generator state as there."""
        dev = self.model._compute_device(x_in, self.betas)
        shape = tuple(int(s) for s in target_shape)
        S = self.sampling_timesteps
        drew_init = init_noise is None
        if drew_init:
            init_noise = torch.randn(shape, device=dev)          # DIFF:275
        if step_noise is None and (drew_init or self.ddim_sampling_eta != 0):
            draws = [torch.randn(shape, device=dev) for _ in range(S - 1)]          # DIFF:293 (the last pair has time_next < 0)
            if self.ddim_sampling_eta != 0:
                step_noise = torch.stack(draws + [torch.zeros(shape, device=dev)], dim=0)   # entry S-1 is never read
        return dev, init_noise, step_noise

    @torch.no_grad()
    def ddim_sample_loop(self, x_in, target_shape, init_noise=None, step_noise=None):
        dev, init_noise, step_noise = self._draw(x_in, target_shape, init_noise, step_noise)
        y0 = self.model._guarded(lambda fb: self._engine(dev, fb), lambda eng: eng.ddim_sample(x_in, init_noise, step_noise),
                                 "ddim_sample_loop")
        return y0.to(x_in.device)

    @torch.no_grad()
    def ddim_sample_loop_ouput_reverse_diffusion(self, x_in, target_shape, init_noise=None, step_noise=None):
        dev, init_noise, step_noise = self._draw(x_in, target_shape, init_noise, step_noise)
        y0, rev, x0s = self.model._guarded(lambda fb: self._engine(dev, fb),
                                           lambda eng: eng.ddim_sample(x_in, init_noise, step_noise, trajectory=True),
                                           "ddim_sample_loop_ouput_reverse_diffusion")
        if self.seq2frame:  # DIFF-S2F:319 records the initial noise as trajectory entry 0
            rev = torch.cat([init_noise.to(rev.device).unsqueeze(-1), rev], dim=-1)
        return y0.to(x_in.device), rev.to(x_in.device), x0s.to(x_in.device)

    def forward_estimate_pose(self, x, target_shape, output_reverse_diffusion_3d=False, **kw):
        if output_reverse_diffusion_3d:
            return self.ddim_sample_loop_ouput_reverse_diffusion(x, target_shape, **kw)
        return self.ddim_sample_loop(x, target_shape, **kw)

    # ------------------------------------------------------------------ forward-process pieces (DIFF:360-419)
    @torch.no_grad()
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        dev = self.model._compute_device(x_start, self.betas)
        return self._engine(dev).q_sample(x_start, t, noise).to(x_start.device)

    @torch.no_grad()
    def get_noisy_pose(self, x_start, num_sample, noise=None):
        t_list = list(np.arange(0, self.num_timesteps, self.num_timesteps // num_sample))
        xs = [x_start] if self.seq2frame else []          # DIFF-S2F:383 seeds the list with x_start
        for t_sample in t_list:
            t = torch.full((x_start.shape[0],), int(t_sample), dtype=torch.long, device=x_start.device)
            xs.append(self.q_sample(x_start=x_start, t=t, noise=noise))
        x_diffusion = torch.stack(xs, dim=-1)
        return x_diffusion if self.seq2frame else (x_diffusion, t_list)

    @property
    def loss_fn(self):
        if self.loss_type == 'l1':
            return F.l1_loss
        if self.loss_type == 'l2':
            return F.mse_loss
        raise ValueError(f'invalid loss type {self.loss_type}')

    @torch.no_grad()
    def p_losses(self, x_start, pose_2d, noise=None, t=None):
        """Forward-only weighted loss (DIFF:392-419): q_sample and the denoiser run in the engine; no autograd."""
        b = x_start.shape[0]
        own_t = t is None
        if own_t:
            t = torch.randint(0, self.num_timesteps, (b,), device=x_start.device).long()
        if noise is None:
            noise = torch.randn_like(x_start)
        self.loss_fn                                                                     # (ValueError on an unknown loss_type, DIFF:375)
        dev = self.model._compute_device(pose_2d, self.betas)

        def run(eng):
            x_noisy = eng.q_sample(x_start, t, noise, check_t=not own_t)
            model_out = eng.denoise(pose_2d, x_noisy, t)                                 # y broadcast over T for seq2frame
            # the variable loss weight 1 + k_t, its clamp and the loss itself (DIFF:411-418) are one engine kernel (d3d_weighted_loss)
            return eng.weighted_loss(model_out, x_start, t, self.loss_type, self.clipLoss, check_t=False)
        return self.model._guarded(lambda fb: self._engine(dev, fb), run, "p_losses").to(x_start.device)

    # ------------------------------------------------------------------ forward (DIFF:421-449)
    def forward(self, clean_3d_pose, noisy_2d_pose, noise=None, output_reverse_diffusion_3d=False, output_loss=True,
                repeat_n=1, init_noise=None, step_noise=None):
        if self.training:
            warnings.warn("diff3dhpe_amd is an inference engine: the training-mode loss carries no gradient", stacklevel=2)
            return self.p_losses(clean_3d_pose, noisy_2d_pose, noise), None
        loss_pose = self.p_losses(clean_3d_pose, noisy_2d_pose, noise) if output_loss else None
        b, f, p, c = clean_3d_pose.shape
        in_dev = noisy_2d_pose.device
        if repeat_n != 1:   # .repeat(repeat_n, 1, 1, 1) (DIFF:433) and the mean over the hypotheses below: engine kernels
            noisy_2d_pose = repeat_batch(noisy_2d_pose.to(self.model._compute_device(noisy_2d_pose, self.betas)), repeat_n)
        target_shape = list(clean_3d_pose.shape)
        target_shape[0] = target_shape[0] * repeat_n
        res = self.forward_estimate_pose(noisy_2d_pose, target_shape=target_shape,
                                         output_reverse_diffusion_3d=output_reverse_diffusion_3d,
                                         init_noise=init_noise, step_noise=step_noise)
        if output_reverse_diffusion_3d:
            pred, rev, x0s = res
            return loss_pose, hypothesis_mean(pred, repeat_n).to(in_dev), rev.to(in_dev), x0s.to(in_dev)   # DIFF:441
        return loss_pose, hypothesis_mean(res, repeat_n).to(in_dev)                      # DIFF:448
