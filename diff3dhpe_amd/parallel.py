"""Multi-GPU host logic: one process per GPU, batch sharded on dim 0, ONE exchange after the loop.

Replaces the reference's single-process nn.DataParallel (RUN:216-218: per-call weight broadcast + scatter + gather).
Weights stay resident per rank; samples are independent through all DDIM steps (no cross-sample op in DIFF:262-300 or
the denoiser), so the only collective is an all-gather of the predicted sequences (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests) feeding the MPJPE reduction (RUN:602-606, LOSS:15-22).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of dim 0; the first B % world ranks take one extra row (ragged batches allowed)."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def all_gather_pred(pred_local: torch.Tensor, B_total: int, force: bool = False) -> torch.Tensor:
    """Assemble (B_total, ...) on every rank from per-rank shards produced with shard_bounds().
    force: run the collective on a one-rank group too (bench.py with D3D_FORCE_DIST=1: RCCL path on a one-GPU box)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return pred_local
    world, rank = dist.get_world_size(), dist.get_rank()
    if pred_local.is_cuda and dist.get_backend() == "gloo":
        # functional-test mode only (several ranks sharing one GPU cannot form an RCCL communicator): stage through the host
        return all_gather_pred(pred_local.cpu(), B_total).to(pred_local.device)
    sizes = [shard_bounds(B_total, r, world)[1] - shard_bounds(B_total, r, world)[0] for r in range(world)]
    tail = tuple(pred_local.shape[1:])
    if len(set(sizes)) == 1:
        out = torch.empty((B_total,) + tail, dtype=pred_local.dtype, device=pred_local.device)
        dist.all_gather_into_tensor(out, pred_local.contiguous())
        return out
    pad = max(sizes)
    buf = torch.zeros((pad,) + tail, dtype=pred_local.dtype, device=pred_local.device)
    buf[: pred_local.shape[0]] = pred_local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def reduce_sums(sum_err: float, count: int, device) -> Tuple[float, int]:
    """Sum (error, joint count) pairs over ranks -- the frame-weighted running mean of RUN:602-606."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return sum_err, count
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([sum_err, float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0]), int(round(float(t[1])))
