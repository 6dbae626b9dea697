"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product path.

A from-scratch eager PyTorch-CPU restatement of the reference's DDIM sampling
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; ``diff3dhpe_amd`` must not.

Parity pin: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
/root/reference in the build container and stores its inputs/outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors (the reference itself ships no tests or fixtures,
SURVEY.md section 8c).

The restatement is written as plain functions over a flat ``state_dict`` (no
``nn.Module`` graph) but deliberately keeps the reference's *operation
sequence* -- materialised identity, ``repeat``, rearrange copies, unfused
LayerNorm -- so that timing it on the GPU node's host cores is a faithful CPU
baseline.

Reference files (all under /root/reference/common/):
  DIFF  = conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames.py
  DIFF-S2F = conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames.py
  S2S   = nets/model_conditional_diffusion_mixste_s2s_grand_linLift.py
  S2F   = nets/model_conditional_diffusion_mixste_s2f_grand_linLift.py
  LOSS  = loss.py
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- schedules (DIFF:52-94)

def beta_schedule(name: str, timesteps: int) -> Tensor:
    """fp64 beta table. DIFF:52-55 (linear), :58-68 (cosine), :70-81 (logcosine); ValueError DIFF:127."""
    if name == "linear":
        return torch.linspace(0.0001, 0.02, timesteps, dtype=torch.float64)
    if name in ("cosine", "logcosine"):
        s = 0.008
        if name == "cosine":
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        else:
            x = torch.logspace(0, 2, timesteps + 1, dtype=torch.float64)
            ac = torch.cos(((x / 1e-1 / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = 1 - (ac[1:] / ac[:-1])
        return torch.clip(betas, 0, 0.999)
    raise ValueError(f"unknown beta schedule {name}")


def diffusion_tables(name: str, timesteps: int, p2_gamma: float = 0.0, p2_k: float = 1.0) -> Dict[str, Tensor]:
    """The 14 fp32 buffers GaussianDiffusion registers (DIFF:130-183), computed in fp64 then cast."""
    betas = beta_schedule(name, timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    t64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_recip_alphas": torch.sqrt(1.0 / alphas),
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": torch.log(pv.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "p2_loss_weight": (p2_k + ac / (1 - ac)) ** -p2_gamma,
    }
    return {k: v.to(torch.float32) for k, v in t64.items()}


def ddim_times(num_timesteps: int, sampling_timesteps: int) -> List[int]:
    """Reversed integer timestep list, bit-exact with DIFF:270-272 (fp32 linspace then trunc)."""
    times = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1)
    return list(reversed(times.int().tolist()))


def ddim_times_scalar(num_timesteps: int, sampling_timesteps: int) -> List[int]:
    """Scalar restatement of torch's two-sided fp32 linspace (SURVEY.md section 7 'Bit-exact schedule').

    step=(end-start)/(steps-1) in fp32; i < steps/2 ? start + i*step : end - (steps-1-i)*step; trunc to int.
    This is the algorithm the C ABI (d3d_ddim_times) implements; pinned against ddim_times() for S in [1,1000].
    """
    import numpy as np
    f = np.float32
    steps = sampling_timesteps + 1
    start, end = f(-1.0), f(num_timesteps - 1)
    step = f((end - start) / f(steps - 1))
    half = steps // 2
    out = []
    for i in range(steps):
        if i < half:
            v = f(start + f(f(i) * step))
        else:
            v = f(end - f(f(steps - 1 - i) * step))
        out.append(int(v))  # trunc toward zero
    return list(reversed(out))


# --------------------------------------------------------------------------- operand-rounding emulation (bf16 engine mode)
# SURVEY.md section 8(d) "Parity gates": the bf16 engine cannot meet the fp32 gate (bf16 operands alone cost ~4e-2 max-abs,
# Appendix B), so it is gated against THIS restatement run with the same roundings: the operands of the matrix-core products of
# a block -- the qkv / proj / fc1 / fc2 Linears (x and W) and the two attention products (q, k and softmax - I, v) -- rounded to
# bf16 (round-to-nearest-even), everything else fp32: accumulation, bias, residual stream, LayerNorm, softmax, GELU, time
# embedding, fusion layer, head, DDIM update.  Off by default (the fp32 restatement pinned against the reference).
_OPERAND_DTYPE = None


class operand_rounding:
    """``with operand_rounding(torch.bfloat16): ...`` -- the block GEMM / attention operands are rounded to that dtype."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _OPERAND_DTYPE
        self._prev, _OPERAND_DTYPE = _OPERAND_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global _OPERAND_DTYPE
        _OPERAND_DTYPE = self._prev


def _rnd(x: Tensor) -> Tensor:
    return x if _OPERAND_DTYPE is None else x.to(_OPERAND_DTYPE).to(x.dtype)


def _block_linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return F.linear(_rnd(x), _rnd(w), b)


# --------------------------------------------------------------------------- denoiser (S2S / S2F)

def sinusoid(time: Tensor, dim: int) -> Tensor:
    """S2S:29-36. Always emits fp32 frequencies."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -e)
    arg = time[:, None] * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_trunk(sd: Dict[str, Tensor], time: Tensor, D: int) -> Tensor:
    """S2S:169-174: sinusoid -> Linear -> GELU(erf) -> Linear. (B,) -> (B, 2D)."""
    h = sinusoid(time, D)
    h = F.linear(h, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def grand_attention(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int, qk_scale: Optional[float] = None) -> Tensor:
    """S2S:73-86. x: (G, N, C). (softmax(q k^T * scale) - I) v, then proj. Identity is materialised as in the reference.
    scale = qk_scale or head_dim ** -0.5 (S2S:65); a layer built with qkv_bias=False has no ".qkv.bias" entry (S2S:67)."""
    G, N, C = x.shape
    dh = C // heads
    qkv = _block_linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias")).reshape(G, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (_rnd(q) @ _rnd(k).transpose(-2, -1)) * (qk_scale or dh ** -0.5)
    a = a.softmax(dim=-1)
    eye = torch.eye(N, dtype=a.dtype).view(1, 1, N, N).repeat(G, heads, 1, 1)
    o = (_rnd(a - eye) @ _rnd(v)).transpose(1, 2).reshape(G, N, C)
    return _block_linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def mixste_block(sd: Dict[str, Tensor], p: str, x: Tensor, spatial: bool, temb: Optional[Tensor], heads: int,
                 qk_scale: Optional[float] = None, norm_eps: float = 1e-6) -> Tensor:
    """S2S:111-135 (eval branch). x: (b, f, j, c).  norm_eps: the eps of the constructor's norm_layer (S2S:184: 1e-6 by default)."""
    b, f, j, c = x.shape
    if temb is not None and (p + ".time_mlp.1.weight") in sd:
        te = F.linear(F.silu(temb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
        x = x + te[:, None, None, :]
    if spatial:
        x = x.reshape(b * f, j, c)
    else:
        x = x.permute(0, 2, 1, 3).reshape(b * j, f, c)  # real transpose copy, as einops does
    h = F.layer_norm(x, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], norm_eps)
    x = x + grand_attention(sd, p + ".attn", h, heads, qk_scale)
    h = F.layer_norm(x, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], norm_eps)
    h = _block_linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    h = F.gelu(h)
    h = _block_linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    x = x + h
    if spatial:
        return x.reshape(b, f, j, c)
    return x.reshape(b, j, f, c).permute(0, 2, 1, 3).contiguous()


def forward_denoise(sd: Dict[str, Tensor], x_cat: Tensor, time: Tensor, *, depth: int, heads: int = 8,
                    seq2frame: bool = False, qk_scale: Optional[float] = None, norm_eps: float = 1e-6) -> Tensor:
    """S2S:249-257 / S2F:253-266. x_cat (B,T,J,in+3), time (B,) -> (B,T,J,3) [(B,1,J,3) for seq2frame].

    ``sd`` holds the denoiser tensors without the 'model.' prefix.
    """
    D = sd["fusion_layer.weight"].shape[0]
    x = F.linear(x_cat, sd["fusion_layer.weight"], sd["fusion_layer.bias"])
    temb = time_trunk(sd, time, D) if "time_mlp.1.weight" in sd else None
    b, f, j, _ = x.shape
    for i in range(depth):
        if i == 0:
            x = (x.reshape(b * f, j, D) + sd["Spatial_pos_embed"]).reshape(b, f, j, D)
        x = mixste_block(sd, f"STEblocks.{i}", x, True, temb, heads, qk_scale, norm_eps)
        x = F.layer_norm(x, (D,), sd["Spatial_norm.weight"], sd["Spatial_norm.bias"], norm_eps)
        if i == 0:
            xt = x.permute(0, 2, 1, 3).reshape(b * j, f, D) + sd["Temporal_pos_embed"]
            x = xt.reshape(b, j, f, D).permute(0, 2, 1, 3).contiguous()
        x = mixste_block(sd, f"TTEblocks.{i}", x, False, temb, heads, qk_scale, norm_eps)
        x = F.layer_norm(x, (D,), sd["Temporal_norm.weight"], sd["Temporal_norm.bias"], norm_eps)
    if seq2frame:
        # S2F:261-263: Conv1d(T->1, k=1) over view(b, f, J*D)
        x = F.conv1d(x.reshape(b, f, j * D), sd["weighted_mean.weight"], sd["weighted_mean.bias"]).reshape(b, 1, j, D)
    x = F.layer_norm(x, (D,), sd["head.0.weight"], sd["head.0.bias"], 1e-5)
    return F.linear(x, sd["head.1.weight"], sd["head.1.bias"])


# --------------------------------------------------------------------------- DDIM loop (DIFF:250-300)

@torch.no_grad()
def ddim_sample_loop(sd: Dict[str, Tensor], tables: Dict[str, Tensor], x2d: Tensor, init_noise: Tensor, *,
                     num_timesteps: int, sampling_timesteps: int, depth: int, heads: int = 8, eta: float = 0.0,
                     clip_denoised: bool = True, seq2frame: bool = False, step_noise: Optional[List[Tensor]] = None,
                     return_trajectory: bool = False):
    """DIFF:262-300 (and :303-347 with return_trajectory; DIFF-S2F:263-300 for seq2frame).

    ``init_noise`` replaces the reference's torch.randn(target_shape); ``step_noise[i]`` replaces the
    per-step randn_like (only matters when eta > 0).  Keeps the DIFF:296 ``alpha * x_start`` form.
    """
    ac = tables["alphas_cumprod"]
    somac = tables["sqrt_one_minus_alphas_cumprod"]
    times = ddim_times(num_timesteps, sampling_timesteps)
    y = init_noise.clone()
    f = x2d.shape[1]
    rev, x0s = [], []
    if seq2frame:
        rev.append(y)  # DIFF-S2F:319 records the initial noise as the first trajectory entry (S+1 entries)
    for idx, (t, t_next) in enumerate(zip(times[:-1], times[1:])):
        tvec = torch.full((y.shape[0],), t, dtype=torch.long)
        y_in = y.repeat(1, f, 1, 1) if seq2frame else y
        x0 = forward_denoise(sd, torch.cat([x2d, y_in], dim=-1), tvec, depth=depth, heads=heads, seq2frame=seq2frame)
        if clip_denoised:
            x0 = torch.clamp(x0, min=-1.0, max=1.0)
        x0s.append(x0)
        if t_next < 0:
            y = x0
            rev.append(y)
            continue
        a = ac[t].view(-1, 1, 1, 1)
        an = ac[t_next].view(-1, 1, 1, 1)
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
        c = (1 - an - sigma ** 2).sqrt()
        nz = step_noise[idx] if step_noise is not None else torch.zeros_like(y)
        y = x0 * an.sqrt() + c * ((y - a * x0) / somac[t].view(-1, 1, 1, 1)) + sigma * nz
        rev.append(y)
    if return_trajectory:
        return y, torch.stack(rev, dim=-1), torch.stack(x0s, dim=-1)
    return y


# --------------------------------------------------------------------------- training-side pieces ("next" rows)

def q_sample(tables: Dict[str, Tensor], x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """DIFF:360-366 with extract() DIFF:21-24."""
    shp = (t.shape[0],) + (1,) * (x_start.dim() - 1)
    return (tables["sqrt_alphas_cumprod"].gather(-1, t).reshape(shp) * x_start
            + tables["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(shp) * noise)


@torch.no_grad()
def p_losses(sd, tables, x_start: Tensor, pose_2d: Tensor, t: Tensor, noise: Tensor, *, depth: int, heads: int = 8,
             loss_type: str = "l2", clip_loss: bool = False, seq2frame: bool = False) -> Tensor:
    """DIFF:392-419 with the random draws (t, noise) supplied by the caller."""
    f = pose_2d.shape[1]
    x_noisy = q_sample(tables, x_start, t, noise)
    y_in = x_noisy.repeat(1, f, 1, 1) if seq2frame else x_noisy
    out = forward_denoise(sd, torch.cat([pose_2d, y_in], dim=-1), t, depth=depth, heads=heads, seq2frame=seq2frame)
    coef = 1.0 + tables["alphas_cumprod"][t].view(-1, 1, 1, 1) / tables["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    if clip_loss:
        coef = torch.clamp(coef, max=3.0)
    fn = F.l1_loss if loss_type == "l1" else F.mse_loss
    return fn(out, x_start, reduction="none") * coef


# --------------------------------------------------------------------------- evaluate() math (RUN:583-590, LOSS:15-27)

H36M_JOINTS_LEFT = [4, 5, 6, 11, 12, 13]
H36M_JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]


def mpjpe(pred: Tensor, target: Tensor) -> Tensor:
    """LOSS:15-22 (reduce='mean')."""
    assert pred.shape == target.shape
    return torch.mean(torch.norm(pred - target, dim=target.dim() - 1))


def merge_flip_tta(pred: Tensor, pred_flip: Tensor, scale: float, target_mask: Tensor,
                   joints_left=H36M_JOINTS_LEFT, joints_right=H36M_JOINTS_RIGHT) -> Tensor:
    """RUN:583-590: un-flip, average, de-normalise, flatten frames, keep masked frames. -> (Nvalid,1,J,3)."""
    pf = pred_flip.clone()
    pf[:, :, :, 0] *= -1
    pf[:, :, joints_left + joints_right] = pf[:, :, joints_right + joints_left]
    p = (pred + pf) / 2.0
    p = p * scale
    J = p.shape[2]
    p = p.reshape(-1, J, 3)
    return p[target_mask.reshape(-1) == True, :, :].unsqueeze(1)  # noqa: E712


# --------------------------------------------------------------------------- evaluate()'s other three protocols (RUN:602-614, LOSS:43-93, 132-142)

def n_mpjpe(pred: Tensor, target: Tensor) -> Tensor:
    """LOSS:83-93 (Protocol #3): one scale per frame, <target, pred> / <pred, pred> over the frame's joints, then mpjpe.  pred / target:
    (N, 1, J, 3) as evaluate() passes them (RUN:603)."""
    assert pred.shape == target.shape and pred.dim() == 4
    den = torch.mean(torch.sum(pred ** 2, dim=3, keepdim=True), dim=2, keepdim=True)
    num = torch.mean(torch.sum(target * pred, dim=3, keepdim=True), dim=2, keepdim=True)
    return mpjpe((num / den) * pred, target)


def p_mpjpe(pred, target):
    """LOSS:43-81 (Protocol #2) on numpy arrays (N, J, 3), in their own dtype (float32 from evaluate(), RUN:608-611): per frame the
    similarity transform (scale a, rotation R without reflection, translation t) that best maps the prediction onto the target --
    centred and norm-scaled point sets, H = X0^T Y0 = U S V^T, R = V U^T with the last singular direction flipped when det R < 0 --, then
    the mean joint distance of the aligned prediction."""
    import numpy as np
    assert pred.shape == target.shape and pred.ndim == 3
    mu_t, mu_p = target.mean(axis=1, keepdims=True), pred.mean(axis=1, keepdims=True)
    t0, p0 = target - mu_t, pred - mu_p
    nt = np.sqrt((t0 ** 2).sum(axis=(1, 2), keepdims=True))
    npd = np.sqrt((p0 ** 2).sum(axis=(1, 2), keepdims=True))
    t0 = t0 / nt
    p0 = p0 / npd
    U, sv, Vt = np.linalg.svd(np.matmul(t0.transpose(0, 2, 1), p0))
    V = Vt.transpose(0, 2, 1)
    sign = np.sign(np.expand_dims(np.linalg.det(np.matmul(V, U.transpose(0, 2, 1))), axis=1))
    V = V.copy()
    sv = sv.copy()
    V[:, :, -1] *= sign
    sv[:, -1] *= sign.flatten()
    R = np.matmul(V, U.transpose(0, 2, 1))
    a = np.expand_dims(sv.sum(axis=1, keepdims=True), axis=2) * nt / npd
    shift = mu_t - a * np.matmul(mu_p, R)
    aligned = a * np.matmul(pred, R) + shift
    return np.mean(np.linalg.norm(aligned - target, axis=2))


def mean_velocity_error(pred, target):
    """LOSS:132-142 (MPJVE) on numpy arrays (N, J, 3): first differences along axis 0 -- the batch's kept frames in their flattened order,
    window and sequence boundaries included --, mean joint distance of the difference.  N = 1: the mean of an empty array (nan), as there."""
    import numpy as np
    assert pred.shape == target.shape
    return np.mean(np.linalg.norm(np.diff(pred, axis=0) - np.diff(target, axis=0), axis=target.ndim - 1))


def protocol_sums(pred: Tensor, target: Tensor):
    """What one batch adds to evaluate()'s four running sums (RUN:602-614): pred / target (N, 1, J, 3) merged, de-normalised, masked.
    Returns (N, N * mpjpe, N * p_mpjpe, N * n_mpjpe, N * mean_velocity_error) as Python floats."""
    n = pred.shape[0] * pred.shape[1]
    e1 = mpjpe(pred, target).item()
    e3 = n_mpjpe(pred, target).item()
    pn = pred.cpu().numpy().reshape(-1, pred.shape[-2], pred.shape[-1])
    tn = target.cpu().numpy().reshape(-1, target.shape[-2], target.shape[-1])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # (N = 1: numpy's "mean of empty slice")
        e2, ev = float(p_mpjpe(pn, tn)), float(mean_velocity_error(pn, tn))
    return n, n * e1, n * e2, n * e3, n * ev


# --------------------------------------------------------------------------- eval windows of a whole sequence (GEN:27-48, 247-271)

def chunk_index(n_frames: int, T: int):
    """Window table of ChunkedGenerator(out_all=True, pad=0) for one sequence: non-overlapping T-frame chunks, the last
    chunk shifted back to end at the last frame; returns (start_index (nc,), target_mask (nc, T) bool).
    GEN:31-44 builds the bounds; GEN:263-271 masks the frames of the last chunk that the previous chunk already covered.
    A sequence shorter than T is edge-padded on the left (GEN:255-260) and fully unmasked (LOAD:270-271)."""
    import numpy as np
    nc = (n_frames + T - 1) // T
    starts = np.arange(nc) * T
    starts[-1] = n_frames - T
    mask = np.ones((nc, T), dtype=bool)
    n_unused = nc * T - n_frames          # == start_3d - start_target_3d of the last chunk
    if n_frames >= T and n_unused > 0:
        mask[-1, :n_unused] = False
    return starts, mask


def gather_windows(seq: Tensor, T: int, flip: bool = False, left=None, right=None):
    """(n, J, C) -> (nc, T, J, C) windows (edge-clamped), optionally the horizontally flipped copy (GEN:273-276)."""
    import numpy as np
    n = seq.shape[0]
    starts, mask = chunk_index(n, T)
    idx = np.clip(starts[:, None] + np.arange(T)[None, :], 0, n - 1)
    w = seq[torch.from_numpy(idx)].clone()
    if flip:
        w[..., 0] *= -1
        w[:, :, list(left) + list(right)] = w[:, :, list(right) + list(left)]
    return w, torch.from_numpy(mask)


# --------------------------------------------------------------------------- seq2frame windows (GEN:402-420, 492-552; LOAD:312-316)

def chunk_index_s2f(n_frames: int, stride: int = 1):
    """Pair table of ChunkedGenerator_3dhp(out_all=False) for one sequence (GEN:402-420): n_chunks = ceil(n / stride) chunks of
    `stride` target frames, centred by offset = (n_chunks * stride - n) // 2; returns the (start_3d, end_3d) bounds."""
    import numpy as np
    n_chunks = (n_frames + stride - 1) // stride
    offset = (n_chunks * stride - n_frames) // 2
    bounds = np.arange(n_chunks + 1) * stride - offset
    return bounds[:-1], bounds[1:]


def gather_windows_s2f(seq2d: Tensor, seq3d: Tensor, valid, T: int, flip: bool = False, left=None, right=None, stride: int = 1):
    """Seq2frame evaluation items of one sequence (get_batch_seq2frame, GEN:492-552, test split): per chunk the 2D frames
    [start - pad, end + pad) with pad = (T - 1) // 2, edge-padded (np.pad 'edge'); the 3D target frames [start, end) (edge-padded
    when the centred chunk table reaches beyond the sequence); target_mask = valid[start:end] as bool (None without `valid`).
    -> (windows_2d (nc, stride + 2 pad, J, C), targets_3d (nc, stride, J, 3), mask (nc, stride) bool | None)."""
    import numpy as np
    n = seq2d.shape[0]
    pad = (T - 1) // 2
    s3, e3 = chunk_index_s2f(n, stride)
    idx2 = np.clip((s3 - pad)[:, None] + np.arange(stride + 2 * pad)[None, :], 0, n - 1)
    idx3 = np.clip(s3[:, None] + np.arange(stride)[None, :], 0, n - 1)
    w = seq2d[torch.from_numpy(idx2)].clone()
    g = seq3d[torch.from_numpy(idx3)].clone()
    if flip:
        w[..., 0] *= -1
        w[:, :, list(left) + list(right)] = w[:, :, list(right) + list(left)]
    m = None
    if valid is not None:
        m = torch.from_numpy(np.asarray(valid).reshape(n, -1)[:, 0][idx3].astype(bool))
    return w, g, m


def evaluate_batch(sd, tables, x2d: Tensor, x2d_flip: Tensor, gt: Tensor, target_mask: Tensor, noise: Tensor, noise_flip: Tensor, *,
                   scale: float, depth: int, sampling_timesteps: int, num_timesteps: int = 1000, seq2frame: bool = False,
                   joints_left=H36M_JOINTS_LEFT, joints_right=H36M_JOINTS_RIGHT):
    """One batch of the reference's evaluate() (RUN:575-606 / RUN3DHP:510-533): two samplings (normal + flipped 2D input), un-flip /
    average / de-normalise / mask, MPJPE.  -> (mpjpe over the valid frames, number of valid frames)."""
    kw = dict(num_timesteps=num_timesteps, sampling_timesteps=sampling_timesteps, depth=depth, seq2frame=seq2frame)
    p = ddim_sample_loop(sd, tables, x2d, noise, **kw)
    pf = ddim_sample_loop(sd, tables, x2d_flip, noise_flip, **kw)
    merged = merge_flip_tta(p, pf, scale, target_mask, joints_left, joints_right)
    J = gt.shape[2]
    g = gt.reshape(-1, J, 3)[target_mask.reshape(-1) == True, :, :].unsqueeze(1)  # noqa: E712
    return mpjpe(merged, g), g.shape[0]
