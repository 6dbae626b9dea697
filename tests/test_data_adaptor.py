"""Dataset adaptor (diff3dhpe_amd/data.py) against the fixture the reference's load_Dataset(..., 'test') + ChunkedGenerator
produced on the synthetic H36M-shaped data set (oracle/gen_golden.py::gen_dataset; synthetic cameras).  Host-side index and
coordinate math: bit-exact.  No kernel is launched here."""
import numpy as np
import torch

from conftest import gold
from diff3dhpe_amd.data import EvalData, MocapMeta, load_eval_npz, normalize_screen_coordinates, world_to_camera
from diff3dhpe_amd.synth import synth_mocap, write_synth_mocap, SYNTH_JOINTS_LEFT, SYNTH_JOINTS_RIGHT


def _data(T, **kw):
    pos, cams, kp, meta = synth_mocap(0)
    return EvalData(MocapMeta(pos, cams, SYNTH_JOINTS_LEFT, SYNTH_JOINTS_RIGHT), kp, meta["keypoints_symmetry"], ["S9", "S11"], T, **kw)


def test_windows_equal_the_reference_loader_bit_for_bit():
    g = gold("dataset_eval")
    for T in (27, 9):
        ed = _data(T)
        items = list(ed.items())
        assert len(items) == len(ed) == g[f"T{T}/target_mask"].shape[0]
        assert np.float32(ed.scale) == g[f"T{T}/scale"]
        assert np.array_equal(np.stack([it["target_mask"] for it in items]), g[f"T{T}/target_mask"])
        for nm in ("inputs_3d", "inputs_3d_norm", "inputs_2d", "inputs_2d_flip"):
            arr = np.stack([it[nm] for it in items])
            assert arr.dtype == np.float32
            wts = np.arange(1, arr.size + 1, dtype=np.float64).reshape(arr.shape) % 9973.0
            assert np.float64((arr.astype(np.float64) * wts).sum()) == g[f"T{T}/{nm}_checksum"], (T, nm)
            if T == 27:
                assert np.array_equal(arr, g[f"T{T}/{nm}"]), nm


def test_batches_follow_dataloader_order_and_filters(tmp_path):
    ed = _data(27)
    bs = list(ed.batches(10))
    assert [b["inputs_2d"].shape[0] for b in bs] == [10, 10, 4] and bs[0]["target_mask"].dtype == torch.bool
    allw = torch.cat([b["inputs_2d"] for b in bs])
    assert torch.equal(allw, torch.from_numpy(np.stack([it["inputs_2d"] for it in ed.items()])))
    # windows of a sequence shorter than T are edge-padded and fully valid; the shifted last window masks its overlap
    short = [it for it in ed.items() if it["key"][1] == "Wait"]          # 9 frames < 27
    assert len(short) == 2 and all(it["target_mask"].all() for it in short)
    assert np.array_equal(short[0]["inputs_2d"][0], short[0]["inputs_2d"][17])           # left edge padding (GEN:259-262)
    assert not np.array_equal(short[0]["inputs_2d"][18], short[0]["inputs_2d"][26])
    walk = [it for it in ed.items() if it["key"] == ("S9", "Walk 1", 0)]   # 70 frames: windows 0-26, 27-53, 43-69
    assert [int(it["target_mask"].sum()) for it in walk] == [27, 27, 16]
    # action filter by prefix (LOAD:185-193), downsampling (LOAD:228-231), the npz route
    assert {it["key"][1] for it in _data(27, actions=["Walk"]).items()} == {"Walk 1"}
    assert len(_data(27, downsample=2)) < len(ed)
    pos, cams, kp, meta = write_synth_mocap(str(tmp_path), seed=0)
    ed2 = load_eval_npz(MocapMeta(pos, cams, SYNTH_JOINTS_LEFT, SYNTH_JOINTS_RIGHT), str(tmp_path), "h36m", "synth", ["S9", "S11"], 27)
    assert len(ed2) == len(ed) and ed2.scale == ed.scale


def test_coordinate_helpers():
    x = np.array([[0.0, 0.0], [1000.0, 1002.0], [500.0, 501.0]], dtype=np.float32)
    n = normalize_screen_coordinates(x, w=1000, h=1002)
    assert np.allclose(n, [[-1.0, -1.002], [1.0, 1.002], [0.0, 0.0]], atol=1e-6)
    q = np.array([1.0, 0.0, 0.0, 0.0], dtype=np.float32)          # identity rotation: world_to_camera = X - t
    X = np.arange(12, dtype=np.float32).reshape(2, 2, 3)
    t = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    assert np.allclose(world_to_camera(X, q, t), X - t)
