"""Deterministic weight and input synthesis.

No dataset or checkpoint is reachable from the build or GPU boxes, so every
test and benchmark draws weights from a counter-based hash (splitmix64 of
(seed, crc32(name), flat index)) and inputs from ``np.random.RandomState`` --
both bit-reproducible on any host, so a 175 MB weight blob never has to be
committed or shipped.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .spec import DenoiserConfig, denoiser_param_spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, numel: int, seed: int) -> np.ndarray:
    """``numel`` float64 values in [-1, 1), a pure function of (name, index, seed)."""
    key = np.uint64((int(seed) & 0xFFFFFFFF) << 32 | (zlib.crc32(name.encode()) & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        ctr = _splitmix64(np.full(numel, key, dtype=np.uint64)) + np.arange(numel, dtype=np.uint64)
    bits = _splitmix64(ctr) >> np.uint64(11)  # 53 random bits
    return bits.astype(np.float64) * (2.0 / float(1 << 53)) - 1.0


def synth_param(name: str, shape, kind: str, fan_in: int, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    u = hash_uniform(name, n, seed)
    if kind in ("linear_w", "linear_b"):
        v = u / np.sqrt(float(fan_in))
    elif kind == "ln_w":
        v = 1.0 + 0.1 * u
    elif kind == "ln_b":
        v = 0.1 * u
    elif kind == "pos":
        v = 0.02 * u
    else:
        raise ValueError(f"unknown init kind {kind}")
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(cfg: DenoiserConfig, seed: int = 0, prefix: str = "") -> Dict[str, np.ndarray]:
    """Denoiser weights keyed by the reference state-dict names (optionally prefixed, e.g. 'model.')."""
    return {prefix + name: synth_param(name, shape, kind, fan_in, seed)
            for name, shape, kind, fan_in in denoiser_param_spec(cfg)}


def synth_inputs(B: int, T: int, J: int = 17, seed: int = 42, cpn_jitter: bool = True):
    """CPN-style 2D windows, initial noise and a root-centred 3D target.

    Returns dict with x2d (B,T,J,2), noise (B,T,J,3), gt3d (B,T,J,3), all float32.
    """
    rng = np.random.RandomState(seed)
    anchor = rng.uniform(-0.5, 0.5, (B, 1, J, 2))
    motion = np.cumsum(rng.normal(0.0, 0.01, (B, T, J, 2)), axis=1)
    jitter = rng.normal(0.0, 0.01, (B, T, J, 2))
    x2d = anchor + motion + (jitter if cpn_jitter else 0.0)
    x2d = np.clip(x2d, -1.0, 1.0).astype(np.float32)
    noise = rng.standard_normal((B, T, J, 3)).astype(np.float32)
    gt = rng.uniform(-1.0, 1.0, (B, T, J, 3))
    gt = (gt - gt[:, :, :1, :]).astype(np.float32)
    return {"x2d": x2d, "noise": noise, "gt3d": gt}
