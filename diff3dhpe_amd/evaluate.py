"""evaluate()-equivalent harness (reference RUN:535-654; the 3DHP runner's run_..._3dhp.py:479-579) on top of the engine, for
synthetic or user-supplied loaders -- sequence-to-sequence and sequence-to-frame (...S2F...: (B, 1, J, 3) targets) models alike.

Per batch it reproduces the reference's data flow: two full DDIM samplings (normal + horizontally flipped 2D input,
RUN:577-582), un-flip/average/de-normalise/mask (RUN:583-590) and the frame-weighted MPJPE running mean
(RUN:602-606, LOSS:15-22) -- the merge and the error reduction run in one HIP kernel (d3d_tta_mpjpe).
With torch.distributed initialised (one process per GPU) each rank samples its shard of the batch and the predicted
sequences are all-gathered (RCCL) before the reduction.
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.distributed as dist

from . import parallel
from .engine import pose_metrics, tta_mpjpe, window_gather, window_gather_s2f

H36M_JOINTS_LEFT = [4, 5, 6, 11, 12, 13]     # after remove_joints (reference common/h36m_dataset.py:20-21,288)
H36M_JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]


def flip_2d(x2d: torch.Tensor, joints_left: Sequence[int], joints_right: Sequence[int]) -> torch.Tensor:
    """Horizontal-flip augmentation of a 2D window (what the reference dataset hands over as inputs_2d_flip)."""
    f = x2d.clone()
    f[..., 0] *= -1
    f[:, :, list(joints_left) + list(joints_right)] = f[:, :, list(joints_right) + list(joints_left)]
    return f


@torch.no_grad()
def evaluate(model_diffusion, batches: Iterable[Dict[str, torch.Tensor]], *, scale: float = 1.0,
             joints_left: Sequence[int] = H36M_JOINTS_LEFT, joints_right: Sequence[int] = H36M_JOINTS_RIGHT,
             test_time_augmentation: bool = True, device: Optional[torch.device] = None, verbose: bool = True,
             output_loss: bool = False, unit_scale: float = 1000.0, all_protocols: bool = True, collect_predictions: bool = False):
    """batches yield dicts with inputs_2d (B,T,J,2), inputs_3d (B,T',J,3) [ground truth in the data set's unit; T' = T, or 1 for a
    seq2frame model], optional inputs_2d_flip, target_mask (B,T') bool, init_noise / init_noise_flip (B,T',J,3), inputs_3d_norm.
    output_loss=True is the 3DHP runner's call shape (run_..._3dhp.py:517-520 leaves forward()'s default): every sampling is preceded
    by the forward-only p_losses on the normalised ground truth (its flipped copy for the flipped input, :497-500) -- the value is
    discarded there and here; it matters for the generator draws it consumes.  unit_scale multiplies the reported error (1000: H36M
    ground truth in metres -> mm; 1: 3DHP ground truth already in mm).  all_protocols (default, as the reference): besides Protocol #1
    the batch's merged prediction goes through d3d_pose_metrics for P-MPJPE, N-MPJPE and MPJVE, each weighted by the batch's kept frames
    (RUN:602-614) -- False keeps the MPJPE-only tail (one kernel, no merged tensor).  Returns a dict with the four errors, frames,
    seconds; as_reference_tuple(result) is evaluate()'s own return value (e1, e2, e3, ev, N, epoch_time).  collect_predictions: the
    merged, de-normalised predictions of the kept frames, batch after batch, come back as result["predictions"] ((N, J, 3) CPU tensor) --
    what the 3DHP runner stores per test sequence for its inference_data.mat (run_..._3dhp.py:542-547)."""
    model_diffusion.eval()
    dev = device or torch.device("cuda", torch.cuda.current_device())
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    tot_err, tot_cnt, secs, frames = 0.0, 0, 0.0, 0
    tot_p, tot_n, tot_v = 0.0, 0.0, 0.0            # running sums of (kept frames of the batch) x (the batch's protocol value), RUN:603-614
    kept_preds = []
    for batch in batches:
        x2d = batch["inputs_2d"]
        gt = batch["inputs_3d"]
        B = x2d.shape[0]
        mask = batch.get("target_mask")
        x2d_f = batch.get("inputs_2d_flip")
        if test_time_augmentation and x2d_f is None:
            x2d_f = flip_2d(x2d, joints_left, joints_right)
        lo, hi = parallel.shard_bounds(B, rank, world)
        sl = slice(lo, hi)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        shape = gt[sl].shape

        def run_batch():
            pred_f = None
            clean = gt[sl].to(dev)
            clean_f = clean
            if output_loss and hi > lo:     # p_losses reads the values: the normalised ground truth, and its mirror image for the flipped input
                clean = (batch["inputs_3d_norm"][sl].to(dev) if batch.get("inputs_3d_norm") is not None else clean / scale)
                clean_f = clean.clone()
                clean_f[..., 0] *= -1
                clean_f[:, :, list(joints_left) + list(joints_right)] = clean_f[:, :, list(joints_right) + list(joints_left)]
            if hi == lo:        # fewer windows than ranks: nothing to sample on this rank (forward() cannot take an empty batch)
                pred = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
                pred_f = pred.clone() if test_time_augmentation else None
            else:
                _, pred = model_diffusion(clean_3d_pose=clean, noisy_2d_pose=x2d[sl].to(dev), output_loss=output_loss,
                                          init_noise=None if batch.get("init_noise") is None else batch["init_noise"][sl])
            if test_time_augmentation and hi > lo:
                _, pred_f = model_diffusion(clean_3d_pose=clean_f, noisy_2d_pose=x2d_f[sl].to(dev), output_loss=output_loss,
                                            init_noise=None if batch.get("init_noise_flip") is None else batch["init_noise_flip"][sl])
            gsl = sl
            if world > 1:   # the ONE exchange step: all-gather the predicted sequences; every rank then reduces the full batch (the
                # reduction is redundant x world: 3 MB per rank at cfg3, one kernel -- cheaper than a second collective for the sums)
                if pred_f is not None:   # the TTA pair travels together: (b, 2, T, J, 3) shards, one collective per batch
                    both = parallel.all_gather_pred(torch.stack([pred, pred_f], dim=1), B)
                    pred, pred_f = both[:, 0], both[:, 1]
                else:
                    pred = parallel.all_gather_pred(pred, B)
                gsl = slice(0, B)
            gtd, md = gt[gsl].to(dev), (None if mask is None else mask[gsl].to(dev))
            if not all_protocols and not collect_predictions:
                return tta_mpjpe(pred, pred_f, gtd, md, scale, list(joints_left), list(joints_right)) + (None, None)   # (reads the two
                # sums back: the batch's one synchronisation)
            err_, cnt_, merged = tta_mpjpe(pred, pred_f, gtd, md, scale, list(joints_left), list(joints_right), want_merged=True)
            kept_p = None
            if collect_predictions:
                flat = merged.reshape(-1, merged.shape[-2], 3)
                kept_p = (flat if md is None else flat[md.reshape(-1).bool()]).cpu()
            return err_, cnt_, (pose_metrics(merged, gtd, md) if all_protocols else None), kept_p

        net = getattr(getattr(model_diffusion, "module", model_diffusion), "model", None)
        if world == 1 and hasattr(net, "deferred_range_checks"):
            # the range guard of the two samplings is read HERE, behind the synchronisation the batch makes anyway (no wait of its
            # own); a flagged batch is repeated on the model's exact-fp32 engine (precision "auto") or raises D3DError ("f16x3")
            rng = torch.cuda.get_rng_state(dev)          # (a repeated batch draws the same noise: the reference's generator sequence)
            with net.deferred_range_checks() as pending:
                err, cnt, extra, kept_p = run_batch()
            if pending.resolve():
                torch.cuda.set_rng_state(rng, dev)
                err, cnt, extra, kept_p = run_batch()
        else:   # several ranks: a rank must know its own flags BEFORE its shard enters the all-gather -- each sampling waits on its ticket
            err, cnt, extra, kept_p = run_batch()
        if kept_p is not None:
            kept_preds.append(kept_p)
        torch.cuda.synchronize(dev)
        secs += time.time() - t0
        tot_err += err
        tot_cnt += cnt
        frames += cnt // shape[2]
        if extra is not None and extra[0] > 0:       # (a batch without a kept frame adds nothing; one kept frame: mpjve nan, as there)
            kept, en, ep, ev = extra
            tot_n += kept * en
            tot_p += kept * ep
            tot_v += kept * ev
    e1 = tot_err / max(tot_cnt, 1) * unit_scale
    nfr = max(frames, 1)
    e2, e3, ev_ = ((tot_p / nfr * unit_scale, tot_n / nfr * unit_scale, tot_v / nfr * unit_scale) if all_protocols else (None, None, None))
    if verbose and rank == 0:
        print('eval_frame:', frames)
        print('inference_time:', secs / 60, 'min')
        print('inference_speed:', frames / max(secs, 1e-9), 'frame/s')
        print('Protocol #1 Error (MPJPE):', e1, 'mm')
        if all_protocols:
            print('Protocol #2 Error (P-MPJPE):', e2, 'mm')
            print('Protocol #3 Error (N-MPJPE):', e3, 'mm')
            print('Velocity Error (MPJVE):', ev_, 'mm')
    out = {"mpjpe_mm": e1, "p_mpjpe_mm": e2, "n_mpjpe_mm": e3, "mpjve_mm": ev_, "frames": frames, "seconds": secs}
    if collect_predictions:
        out["predictions"] = torch.cat(kept_preds) if kept_preds else torch.empty(0)
    return out


def as_reference_tuple(result: Dict[str, float]):
    """evaluate()'s return value in the reference's own shape (RUN:654): (e1, e2, e3, ev, N, epoch_time)."""
    return (result["mpjpe_mm"], result["p_mpjpe_mm"], result["n_mpjpe_mm"], result["mpjve_mm"], result["frames"], result["seconds"])


@torch.no_grad()
def run_evaluation(model_diffusion, data, *, batch_size: int = 1024, action_filter: Optional[Sequence[str]] = None, verbose: bool = True,
                   unit_scale: float = 1000.0, noise_std: float = 0.0, joint_drop_rate: float = 0.0, **evaluate_kw):
    """The runner's per-action evaluation loop (RUN:712-766): for every action name of the data set (`data.action_names()`: the first
    word of the action, RUN:669-682) that starts with one of `action_filter` (None: all) -- one evaluate() over the windows of the
    actions with that PREFIX (`data.batches(batch_size, action_filter=[name])`, RUN:730-736), then the action-wise averages of the four
    protocols and the totals.  noise_std / joint_drop_rate: the runner's --test_extra_noise_std / --test_joint_drop (RUN:731), handed to the
    data set.  data: diff3dhpe_amd.data.EvalData (or anything with action_names(), batches(), scale, joints_left / joints_right).  Returns {"actions": {name: (e1, e2, e3, ev, N, seconds)}, the four "*_mm" action-wise means, "frames", "seconds"}."""
    per = {}
    for name in data.action_names():
        if action_filter is not None and not any(name.startswith(a) for a in action_filter):
            continue
        extra = {k: v for k, v in (("noise_std", noise_std), ("joint_drop_rate", joint_drop_rate)) if v}
        r = evaluate(model_diffusion, data.batches(batch_size, action_filter=[name], **extra), scale=data.scale, joints_left=data.joints_left,
                     joints_right=data.joints_right, verbose=False, unit_scale=unit_scale, **evaluate_kw)
        per[name] = as_reference_tuple(r)
        if verbose:
            print('----' + name + '----')
            print('Protocol #1 Error (MPJPE):', r["mpjpe_mm"], 'mm')
            print('Protocol #2 Error (P-MPJPE):', r["p_mpjpe_mm"], 'mm')
            print('Protocol #3 Error (N-MPJPE):', r["n_mpjpe_mm"], 'mm')
            print('Velocity Error (MPJVE):', r["mpjve_mm"], 'mm')
    frames = sum(v[4] for v in per.values())
    secs = sum(v[5] for v in per.values())
    mean = lambda i: (sum(v[i] for v in per.values()) / len(per)) if per else float("nan")      # np.mean over the actions (RUN:750-753)
    out = {"actions": per, "mpjpe_mm": mean(0), "p_mpjpe_mm": mean(1), "n_mpjpe_mm": mean(2), "mpjve_mm": mean(3), "frames": frames,
           "seconds": secs}
    if verbose:
        print('Total eval_frame:', frames)
        print('Total inference_time:', secs / 60, 'min')
        print('inference_speed:', frames / max(secs, 1e-9), 'frame/s')
        print('Protocol #1   (MPJPE) action-wise average:', round(out["mpjpe_mm"], 1), 'mm')
        print('Protocol #2 (P-MPJPE) action-wise average:', round(out["p_mpjpe_mm"], 1), 'mm')
        print('Protocol #3 (N-MPJPE) action-wise average:', round(out["n_mpjpe_mm"], 1), 'mm')
        print('Velocity      (MPJVE) action-wise average:', round(out["mpjve_mm"], 2), 'mm')
    return out


@torch.no_grad()
def run_evaluation_3dhp(model_diffusion, data, *, batch_size: int = 1024, subjects_test: Optional[Sequence[str]] = None, verbose: bool = True,
                        noise_std: float = 0.0, joint_drop_rate: float = 0.0, output_loss: bool = True, mat_path: Optional[str] = None,
                        **evaluate_kw):
    """The 3DHP runner's evaluation loop (run_..._3dhp.py:593-632): one evaluate() per test sequence (`data.batches(batch_size,
    seq_filter=name)`, :596-604; output_loss=True is forward()'s default, which that runner leaves; errors in the data set's own mm), the
    sequence-wise averages of the four protocols, and the stored predictions -- data_inference[name] = the kept frames' merged predictions
    as a (3, J, N) array (:542-547), written to `mat_path` (inference_data.mat, :631-632: the input of the data set's PCK / AUC scripts)
    when one is given.  data: diff3dhpe_amd.data.EvalData3DHP.  Returns {"sequences": {name: (e1, e2, e3, ev, N, seconds)}, the four
    "*_mm" means, "data_inference": {name: ndarray}}."""
    names = list(subjects_test) if subjects_test is not None else [s[0] for s in data.sequences]
    per, inf = {}, {}
    extra = {k: v for k, v in (("noise_std", noise_std), ("joint_drop_rate", joint_drop_rate)) if v}
    for name in names:
        r = evaluate(model_diffusion, data.batches(batch_size, seq_filter=name, **extra), scale=data.scale, joints_left=data.joints_left,
                     joints_right=data.joints_right, verbose=False, unit_scale=1.0, output_loss=output_loss, collect_predictions=True,
                     **evaluate_kw)
        per[name] = as_reference_tuple(r)
        inf[name] = r["predictions"].permute(2, 1, 0).numpy()
        if verbose:
            print('----' + name + '----')
    mean = lambda i: (sum(v[i] for v in per.values()) / len(per)) if per else float("nan")
    out = {"sequences": per, "mpjpe_mm": mean(0), "p_mpjpe_mm": mean(1), "n_mpjpe_mm": mean(2), "mpjve_mm": mean(3), "data_inference": inf}
    if verbose:
        print('Protocol #1   (MPJPE) action-wise average:', round(out["mpjpe_mm"], 1), 'mm')
        print('Protocol #2 (P-MPJPE) action-wise average:', round(out["p_mpjpe_mm"], 1), 'mm')
        print('Protocol #3 (N-MPJPE) action-wise average:', round(out["n_mpjpe_mm"], 1), 'mm')
        print('Velocity      (MPJVE) action-wise average:', round(out["mpjve_mm"], 2), 'mm')
    if mat_path is not None:
        import scipy.io as scio
        scio.savemat(mat_path, inf)
    return out


@torch.no_grad()
def evaluate_sequence(model_diffusion, poses_2d: torch.Tensor, poses_3d: torch.Tensor, *, num_frames: int, scale: float = 1.0,
                      joints_left: Sequence[int] = H36M_JOINTS_LEFT, joints_right: Sequence[int] = H36M_JOINTS_RIGHT,
                      kps_left: Optional[Sequence[int]] = None, kps_right: Optional[Sequence[int]] = None,
                      test_time_augmentation: bool = True, batch_size: int = 512, device: Optional[torch.device] = None,
                      init_noise=None, init_noise_flip=None, valid: Optional[torch.Tensor] = None, seq2frame: Optional[bool] = None,
                      output_loss: bool = False, unit_scale: float = 1000.0, all_protocols: bool = True):
    """A whole video 2D-in -> MPJPE-out without host round trips: the window table, edge
    padding, target mask and the flipped 2D copy are built on the device, every window goes through the DDIM loop (twice with TTA),
    and merge + masked MPJPE run in one kernel (RUN:583-606).  Two window tables:
      seq2seq   (GEN:27-48, 247-276; LOAD:243-261) non-overlapping T-frame windows, the last one shifted and its overlap masked
      seq2frame (GEN:402-420, 492-552; LOAD:312-316; chosen by the model's class, or seq2frame=True) ONE window per frame f -- the 2D
                frames f - pad .. f + pad, edge-replicated -- with the single 3D frame f as target: (n, 1, J, 3)
    poses_2d (n, J, 2) normalised screen coordinates, poses_3d (n, J, 3) ground truth divided by `scale` upstream or in the data
    set's unit with scale = its normalisation scale; valid (n,) optional per-frame flags ANDed into the mask (3DHP, GEN:627-628)."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    kl = list(kps_left if kps_left is not None else joints_left)
    kr = list(kps_right if kps_right is not None else joints_right)
    if seq2frame is None:
        seq2frame = bool(getattr(getattr(model_diffusion, "module", model_diffusion), "seq2frame", False))
    p2, p3 = poses_2d.to(dev), poses_3d.to(dev)
    n = p2.shape[0]
    if seq2frame:
        x2d = window_gather_s2f(p2, num_frames)
        x2d_f = window_gather_s2f(p2, num_frames, True, kl, kr) if test_time_augmentation else None
        gt = p3.to(torch.float32).reshape(n, 1, p3.shape[1], 3)
        mask = torch.ones((n, 1), dtype=torch.bool, device=dev) if valid is None else valid.to(dev).reshape(n, 1).bool()
    else:
        x2d, mask = window_gather(p2, num_frames)
        gt = window_gather(p3, num_frames, want_mask=False)
        x2d_f = window_gather(p2, num_frames, True, kl, kr, want_mask=False) if test_time_augmentation else None
        if valid is not None:     # the frames' own flags, through the same window table (edge-clamped like the poses)
            vw = window_gather(valid.to(dev).to(torch.float32).reshape(n, 1, 1), num_frames, want_mask=False)
            mask = mask & (vw.reshape(mask.shape) != 0)
    batches = []
    for lo in range(0, x2d.shape[0], batch_size):
        sl = slice(lo, lo + batch_size)
        b = {"inputs_2d": x2d[sl], "inputs_3d": gt[sl], "target_mask": mask[sl]}
        if x2d_f is not None:
            b["inputs_2d_flip"] = x2d_f[sl]
        if init_noise is not None:
            b["init_noise"] = init_noise[sl]
        if init_noise_flip is not None:
            b["init_noise_flip"] = init_noise_flip[sl]
        batches.append(b)
    return evaluate(model_diffusion, batches, scale=scale, joints_left=joints_left, joints_right=joints_right,
                    test_time_augmentation=test_time_augmentation, device=dev, verbose=False, output_loss=output_loss,
                    unit_scale=unit_scale, all_protocols=all_protocols)
