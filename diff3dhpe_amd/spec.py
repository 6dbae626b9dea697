"""Parameter inventory of the MixSTE denoiser (names, shapes, init kind).

The names are the reference's ``state_dict`` keys so that reference checkpoints
load unchanged (reference: common/nets/model_conditional_diffusion_mixste_s2s_grand_linLift.py:139-220,
common/nets/model_conditional_diffusion_mixste_s2f_grand_linLift.py:216-218).

This table is the single source of truth for
  * the host-side ``nn.Module`` mirror (diff3dhpe_amd.nets),
  * the deterministic weight synthesiser (diff3dhpe_amd.synth),
  * the order in which weights are handed to the C ABI (include/d3d.h).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import List, Tuple

S2S_NAME = "ConditionalDiffusionMixSTES2SGRANDLinLift"
S2F_NAME = "ConditionalDiffusionMixSTES2FGRANDLinLift"


@dataclass(frozen=True)
class DenoiserConfig:
    """Constructor arguments that change tensor shapes (S2S:140-142)."""
    num_frame: int = 9
    num_joints: int = 17
    in_chans: int = 2
    embed_dim: int = 32
    depth: int = 4
    num_heads: int = 8
    mlp_ratio: float = 2.0
    with_time_emb: bool = True
    seq2frame: bool = False

    @property
    def mlp_hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def time_dim(self) -> int:
        return self.embed_dim * 2 if self.with_time_emb else 0

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    def as_dict(self):
        return asdict(self)


# kind: 'linear_w' / 'linear_b' (U(+-1/sqrt(fan_in))), 'ln_w' (1+0.1u), 'ln_b' (0.1u), 'pos' (0.02u)
ParamSpec = Tuple[str, Tuple[int, ...], str, int]


def _block(prefix: str, D: int, Dm: int, Dt: int) -> List[ParamSpec]:
    out: List[ParamSpec] = [
        (f"{prefix}.norm1.weight", (D,), "ln_w", 0),
        (f"{prefix}.norm1.bias", (D,), "ln_b", 0),
        (f"{prefix}.attn.qkv.weight", (3 * D, D), "linear_w", D),
        (f"{prefix}.attn.qkv.bias", (3 * D,), "linear_b", D),
        (f"{prefix}.attn.proj.weight", (D, D), "linear_w", D),
        (f"{prefix}.attn.proj.bias", (D,), "linear_b", D),
        (f"{prefix}.norm2.weight", (D,), "ln_w", 0),
        (f"{prefix}.norm2.bias", (D,), "ln_b", 0),
    ]
    if Dt:
        out += [
            (f"{prefix}.time_mlp.1.weight", (D, Dt), "linear_w", Dt),
            (f"{prefix}.time_mlp.1.bias", (D,), "linear_b", Dt),
        ]
    out += [
        (f"{prefix}.mlp.fc1.weight", (Dm, D), "linear_w", D),
        (f"{prefix}.mlp.fc1.bias", (Dm,), "linear_b", D),
        (f"{prefix}.mlp.fc2.weight", (D, Dm), "linear_w", Dm),
        (f"{prefix}.mlp.fc2.bias", (D,), "linear_b", Dm),
    ]
    return out


def denoiser_param_spec(cfg: DenoiserConfig) -> List[ParamSpec]:
    """All learnable tensors of the denoiser in registration order (S2S:160-220)."""
    D, Dm, Dt, T, J = cfg.embed_dim, cfg.mlp_hidden, cfg.time_dim, cfg.num_frame, cfg.num_joints
    cin = cfg.in_chans + 3
    spec: List[ParamSpec] = []
    if Dt:
        spec += [
            ("time_mlp.1.weight", (Dt, D), "linear_w", D),
            ("time_mlp.1.bias", (Dt,), "linear_b", D),
            ("time_mlp.3.weight", (Dt, Dt), "linear_w", Dt),
            ("time_mlp.3.bias", (Dt,), "linear_b", Dt),
        ]
    spec += [
        ("fusion_layer.weight", (D, cin), "linear_w", cin),
        ("fusion_layer.bias", (D,), "linear_b", cin),
        ("Spatial_pos_embed", (1, J, D), "pos", 0),
    ]
    for i in range(cfg.depth):
        spec += _block(f"STEblocks.{i}", D, Dm, Dt)
    spec += [
        ("Spatial_norm.weight", (D,), "ln_w", 0),
        ("Spatial_norm.bias", (D,), "ln_b", 0),
        ("Temporal_pos_embed", (1, T, D), "pos", 0),
    ]
    for i in range(cfg.depth):
        spec += _block(f"TTEblocks.{i}", D, Dm, Dt)
    spec += [
        ("Temporal_norm.weight", (D,), "ln_w", 0),
        ("Temporal_norm.bias", (D,), "ln_b", 0),
        ("head.0.weight", (D,), "ln_w", 0),
        ("head.0.bias", (D,), "ln_b", 0),
        ("head.1.weight", (3, D), "linear_w", D),
        ("head.1.bias", (3,), "linear_b", D),
    ]
    if cfg.seq2frame:
        spec += [
            ("weighted_mean.weight", (1, T, 1), "linear_w", T),
            ("weighted_mean.bias", (1,), "linear_b", T),
        ]
    return spec


def param_count(cfg: DenoiserConfig) -> int:
    n = 0
    for _, shape, _, _ in denoiser_param_spec(cfg):
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
