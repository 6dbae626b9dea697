"""Reference checkpoint files (checkpoint ingest).

The reference saves ``{'epoch', 'best_epoch', 'min_loss', 'min_train_loss', 'lr', 'random_state', 'optimizer',
'model_diffusion'}`` with ``torch.save`` (RUN:451-460); ``model_diffusion`` is the state dict of the
``nn.DataParallel``-wrapped ``GaussianDiffusion``, so every key carries a ``module.`` prefix (RUN:459).  Its loader drops
every key containing ``'alphas'`` and loads the rest with ``strict=False`` (RUN:226-235), so that the schedule tables of
the run-time ``timesteps`` / ``beta_schedule`` win over the stored ones.  ``load_checkpoint`` does exactly that for the
engine's ``GaussianDiffusion`` (bare or wrapped in ``nn.DataParallel``), ``save_checkpoint`` writes the same layout.
Host-side file handling only: no tensor math happens here.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping

import torch
from torch import nn


def reference_state_dict(checkpoint: Mapping[str, Any]) -> Dict[str, torch.Tensor]:
    """``checkpoint['model_diffusion']`` (or a bare state dict) -> keys without ``module.``, ``'alphas'`` keys dropped."""
    sd = checkpoint["model_diffusion"] if "model_diffusion" in checkpoint else checkpoint
    out = {}
    for k, v in sd.items():
        if "alphas" in k:                      # RUN:228-231
            continue
        out[k[len("module."):] if k.startswith("module.") else k] = v
    return out


def load_checkpoint(model_diffusion: nn.Module, path: str, map_location="cpu") -> Dict[str, Any]:
    """Load a reference ``*.bin`` into ``model_diffusion``; returns the checkpoint's metadata (everything but the weights)
    plus ``missing_keys`` / ``unexpected_keys`` of the non-strict load."""
    try:
        ck = torch.load(path, map_location=map_location, weights_only=False)
    except TypeError:                          # older torch without weights_only
        ck = torch.load(path, map_location=map_location)
    target = model_diffusion.module if isinstance(model_diffusion, nn.DataParallel) else model_diffusion
    res = target.load_state_dict(reference_state_dict(ck), strict=False)
    meta = {k: v for k, v in ck.items() if k != "model_diffusion"} if "model_diffusion" in ck else {}
    meta["missing_keys"], meta["unexpected_keys"] = list(res.missing_keys), list(res.unexpected_keys)
    return meta


def save_checkpoint(model_diffusion: nn.Module, path: str, **meta) -> None:
    """Write the reference's on-disk layout (keys prefixed with ``module.`` as if saved from ``nn.DataParallel``)."""
    target = model_diffusion.module if isinstance(model_diffusion, nn.DataParallel) else model_diffusion
    sd = {"module." + k: v.detach().cpu() for k, v in target.state_dict().items()}
    blob = {"epoch": 0, "best_epoch": 0, "min_loss": float("inf"), "min_train_loss": float("inf"), "lr": 0.0,
            "random_state": None, "optimizer": None}
    blob.update(meta)
    blob["model_diffusion"] = sd
    torch.save(blob, path)
