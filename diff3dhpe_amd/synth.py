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


def _trainedlike_param(name: str, shape, kind: str, fan_in: int, seed: int) -> np.ndarray:
    """Second weight family, "trained-like" statistics (the reference's checkpoints cannot be fetched): heavy-tailed Linear
    weights -- the U(+-1/sqrt(fan_in)) bulk plus a few entries per matrix with |w| in [4, 8] --, LayerNorm gains spread
    log-uniformly over [0.05, 6] with biases up to +-3 gamma, position embeddings of order 1, biases of order 0.5.  It drives the
    F16X3 planes (fp16 exponent range, one-pass LayerNorm statistics) and the softmax (sharp logits) far harder than the
    default family does."""
    n = int(np.prod(shape))
    u = hash_uniform(name, n, seed)
    if kind == "linear_w":
        v = u / np.sqrt(float(fan_in))
        if n >= 4096:        # a few heavy entries (about 1 in 32768, at least 4), positions and values from a second stream
            k = max(4, n // 32768)
            h = hash_uniform(name + "/heavy", 2 * k, seed)
            pos = np.minimum((0.5 * (h[:k] + 1.0) * n).astype(np.int64), n - 1)
            mag = 4.0 + 2.0 * (hash_uniform(name + "/heavy_mag", k, seed) + 1.0)          # [4, 8)
            v[pos] = np.where(h[k:] < 0.0, -mag, mag)
    elif kind == "linear_b":
        v = 0.5 * u
    elif kind == "ln_w":
        v = np.exp(np.log(0.05) + 0.5 * (u + 1.0) * (np.log(6.0) - np.log(0.05)))
    elif kind == "ln_b":
        g = np.exp(np.log(0.05) + 0.5 * (hash_uniform(name.replace(".bias", ".weight"), n, seed) + 1.0) * (np.log(6.0) - np.log(0.05)))
        v = 3.0 * g * u
    elif kind == "pos":
        v = 1.0 * u
    else:
        raise ValueError(f"unknown init kind {kind}")
    return v.astype(np.float32).reshape(shape)


def synth_param(name: str, shape, kind: str, fan_in: int, seed: int, family: str = "uniform") -> np.ndarray:
    if family == "trainedlike":
        return _trainedlike_param(name, shape, kind, fan_in, seed)
    if family != "uniform":
        raise ValueError(f"unknown weight family {family}")
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


def synth_state_dict(cfg: DenoiserConfig, seed: int = 0, prefix: str = "", family: str = "uniform") -> Dict[str, np.ndarray]:
    """Denoiser weights keyed by the reference state-dict names (optionally prefixed, e.g. 'model.').
    family: "uniform" (U(+-1/sqrt(fan_in)), LayerNorm gamma = 1 + 0.1 u) or "trainedlike" (see _trainedlike_param)."""
    return {prefix + name: synth_param(name, shape, kind, fan_in, seed, family)
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


def synth_inputs_rows(lo: int, hi: int, T: int, J: int = 17, seed: int = 42, cpn_jitter: bool = True):
    """Rows lo..hi-1 of an unbounded synthetic batch in which row i is `synth_inputs(1, T, J, seed = hash(seed, i))` -- the recipe of
    SURVEY.md section 8d applied per sequence, so that a rank of a multi-GPU run builds ITS shard in O(shard) host work and any
    sharding of the batch yields the same rows (bench.py)."""
    rows = [synth_inputs(1, T, J, seed=(int(seed) * 1000003 + i) & 0x7FFFFFFF, cpn_jitter=cpn_jitter) for i in range(lo, hi)]
    if not rows:
        return {"x2d": np.zeros((0, T, J, 2), np.float32), "noise": np.zeros((0, T, J, 3), np.float32), "gt3d": np.zeros((0, T, J, 3), np.float32)}
    return {k: np.concatenate([r[k] for r in rows], axis=0) for k in rows[0]}


# ---------------------------------------------------------------------------------------------- synthetic mocap data set
SYNTH_JOINTS_LEFT = [4, 5, 6, 11, 12, 13]      # 17-joint skeleton sides (what the reference's H36M skeleton has after remove_joints)
SYNTH_JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]
SYNTH_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]


def synth_mocap(seed: int = 0):
    """A tiny H36M-SHAPED data set with synthetic cameras (tests/golden/dataset_eval.npz; tests of diff3dhpe_amd.data).

    Returns (positions, cameras, keypoints, metadata):
      positions[subject][action] (n, 17, 3) float32 world coordinates in metres  -> the 'positions_3d' of data_3d_*.npz
      cameras[subject] = list of dicts {orientation (4,) unit quaternion, translation (3,) metres, res_w, res_h, intrinsic (9,)}
      keypoints[subject][action] = list over cameras of (n or n + extra, 17, 2) float32 pixel coordinates -> data_2d_*.npz
      metadata = {'num_joints': 17, 'keypoints_symmetry': [left, right]}
    Sequence lengths exercise a window shorter than T, an exact multiple and ragged tails; one camera track carries extra
    frames (H36M videos do: the loader truncates them)."""
    rng = np.random.RandomState(1000 + seed)
    lengths = {"S9": {"Walk 1": 70, "Sit": 27}, "S11": {"Walk 1": 100, "Eat 2": 55, "Wait": 9}}
    res = [(1000, 1002), (1000, 1000), (1000, 1000), (1000, 1002)]
    positions, cameras, keypoints = {}, {}, {}
    for subject, acts in lengths.items():
        cams = []
        for ci in range(2):
            q = rng.standard_normal(4)
            q = (q / np.linalg.norm(q)).astype(np.float32)
            t = rng.uniform(-3.0, 3.0, 3).astype(np.float32)
            cams.append({"orientation": q, "translation": t, "res_w": res[ci][0], "res_h": res[ci][1],
                         "intrinsic": rng.uniform(-1.0, 1.0, 9).astype(np.float32)})
        cameras[subject] = cams
        positions[subject], keypoints[subject] = {}, {}
        for action, n in acts.items():
            root = np.cumsum(rng.normal(0.0, 0.02, (n, 1, 3)), axis=0) + rng.uniform(-1.0, 1.0, (1, 1, 3))
            body = rng.uniform(-0.9, 0.9, (1, 17, 3)) + np.cumsum(rng.normal(0.0, 0.01, (n, 17, 3)), axis=0)
            positions[subject][action] = (root + body).astype(np.float32)
            tracks = []
            for ci in range(2):
                extra = 3 if (ci == 1 and action == "Walk 1") else 0
                px = rng.uniform(100.0, 900.0, (1, 17, 2)) + np.cumsum(rng.normal(0.0, 3.0, (n + extra, 17, 2)), axis=0)
                tracks.append(px.astype(np.float32))
            keypoints[subject][action] = tracks
    meta = {"num_joints": 17, "keypoints_symmetry": [list(SYNTH_JOINTS_LEFT), list(SYNTH_JOINTS_RIGHT)]}
    return positions, cameras, keypoints, meta


def write_synth_mocap(root: str, seed: int = 0, dataset: str = "h36m", keypoints_name: str = "synth"):
    """Write data_3d_<dataset>.npz and data_2d_<dataset>_<keypoints>.npz in the layout the reference loaders read."""
    import os
    positions, cameras, keypoints, meta = synth_mocap(seed)
    os.makedirs(root, exist_ok=True)
    np.savez(os.path.join(root, f"data_3d_{dataset}.npz"), positions_3d=np.array(positions, dtype=object))
    np.savez(os.path.join(root, f"data_2d_{dataset}_{keypoints_name}.npz"), positions_2d=np.array(keypoints, dtype=object),
             metadata=np.array(meta, dtype=object))
    return positions, cameras, keypoints, meta


def synth_mocap_3dhp(seed: int = 0):
    """A tiny MPI-INF-3DHP-SHAPED data set (tests/golden/dataset_3dhp_eval.npz; tests of diff3dhpe_amd.data.EvalData3DHP), in the
    layout of the data_{train,test}_3dhp.npz files the reference's 3DHP runner reads (common/mpiinf3dhp_dataset.py:21-88):
      test  = {seq: {'data_3d' (n, 17, 3) float64 mm, 'data_2d' (n, 17, 2) float64 pixels, 'valid' (n,) float64 0/1}}
      train = {'S1 Seq1': [{cam: {'data_3d', 'data_2d'}}]}
    TS5 exercises the 1920 x 1080 branch of the screen normalisation; sequence lengths give ragged last windows, an exact
    multiple of the window and invalid frames at both ends and inside."""
    rng = np.random.RandomState(2000 + seed)
    test = {}
    for seq, n in (("TS1", 70), ("TS5", 54), ("TS3", 31)):
        root = np.cumsum(rng.normal(0.0, 20.0, (n, 1, 3)), axis=0) + rng.uniform(-1000.0, 1000.0, (1, 1, 3)) + np.array([0.0, 0.0, 4000.0])
        body = rng.uniform(-800.0, 800.0, (1, 17, 3)) + np.cumsum(rng.normal(0.0, 8.0, (n, 17, 3)), axis=0)
        wh = (1920.0, 1080.0) if seq == "TS5" else (2048.0, 2048.0)
        px = rng.uniform(0.2, 0.8, (1, 17, 2)) * np.array(wh) + np.cumsum(rng.normal(0.0, 3.0, (n, 17, 2)), axis=0)
        valid = (rng.uniform(0.0, 1.0, n) > 0.25).astype(np.float64)
        valid[0], valid[-1] = 0.0, 1.0
        test[seq] = {"data_3d": root + body, "data_2d": px, "valid": valid}
    train = {}
    for seq, n in (("S1 Seq1", 40), ("S2 Seq2", 25)):
        cams = {}
        for cam in ("0", "2"):
            root = np.cumsum(rng.normal(0.0, 20.0, (n, 1, 3)), axis=0) + np.array([0.0, 0.0, 4500.0])
            body = rng.uniform(-900.0, 900.0, (1, 17, 3)) + np.cumsum(rng.normal(0.0, 8.0, (n, 17, 3)), axis=0)
            cams[cam] = {"data_3d": root + body, "data_2d": rng.uniform(0.0, 2048.0, (n, 17, 2))}
        train[seq] = [cams]
    return test, train


def write_synth_3dhp(root: str, seed: int = 0):
    """Write data_test_3dhp.npz / data_train_3dhp.npz as the reference's MPIINF3DHPDataset reads them ('data' = one pickled dict)."""
    import copy
    import os
    test, train = synth_mocap_3dhp(seed)
    os.makedirs(root, exist_ok=True)
    np.savez(os.path.join(root, "data_test_3dhp.npz"), data=np.array(copy.deepcopy(test), dtype=object))
    np.savez(os.path.join(root, "data_train_3dhp.npz"), data=np.array(copy.deepcopy(train), dtype=object))
    return test, train
