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


# ---------------------------------------------------------------------------------------------- MPI-INF-3DHP (BASELINE configs[4])
def _data3dhp(T, out_all, **kw):
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.synth import synth_mocap_3dhp
    test, train = synth_mocap_3dhp(0)
    return EvalData3DHP(test, ["TS1", "TS5"], T, out_all=out_all, train_data=train, **kw), test, train


def test_3dhp_items_equal_the_reference_loader_bit_for_bit():
    """EvalData3DHP against what MPIINF3DHPDataset + load_Dataset_3dhp(split='test') + ChunkedGenerator_3dhp produced on the same
    synthetic files (oracle/gen_golden.py::gen_round5), for the seq2frame table (stride 1, pad 13 / 4) and the seq2seq one."""
    g = gold("dataset_3dhp_eval")
    for tag, T, oa in (("s2f_T27", 27, False), ("s2s_T27", 27, True), ("s2f_T9", 9, False)):
        ed, _, _ = _data3dhp(T, oa)
        items = list(ed.items())
        assert len(items) == len(ed) == g[f"{tag}/target_mask"].shape[0]
        assert np.float32(ed.scale) == g[f"{tag}/scale"]
        assert np.array_equal(np.stack([it["target_mask"] for it in items]), g[f"{tag}/target_mask"])
        for nm in ("inputs_3d", "inputs_3d_norm", "inputs_2d", "inputs_2d_flip"):
            arr = np.stack([it[nm] for it in items])
            assert arr.dtype == np.float32
            wts = np.arange(1, arr.size + 1, dtype=np.float64).reshape(arr.shape) % 9973.0
            assert np.float64((arr.astype(np.float64) * wts).sum()) == g[f"{tag}/{nm}_checksum"], (tag, nm)
            if f"{tag}/{nm}" in g:
                assert np.array_equal(arr, g[f"{tag}/{nm}"]), (tag, nm)
        exp = (1, 17, 3) if not oa else (T, 17, 3)
        assert items[0]["inputs_3d"].shape == exp and items[0]["inputs_2d"].shape == (T, 17, 2)


def test_3dhp_adaptor_details(tmp_path):
    from diff3dhpe_amd.data import EvalData3DHP, load_3dhp
    from diff3dhpe_amd.synth import write_synth_3dhp
    ed, test, train = _data3dhp(27, False)
    # per-sequence data sets (run_evaluation(): seq_filter) and DataLoader batches
    assert ed.num_items("TS1") == 70 and ed.num_items("TS5") == 54 and len(ed) == 124
    bs = list(ed.batches(32, seq_filter="TS1"))
    assert [b["inputs_2d"].shape[0] for b in bs] == [32, 32, 6] and bs[0]["inputs_3d"].shape == (32, 1, 17, 3)
    assert bs[0]["target_mask"].shape == (32, 1) and bs[0]["target_mask"].dtype == torch.bool and not bs[0]["target_mask"][0, 0]
    # edge replication at both ends: window 0 repeats frame 0 thirteen times on the left, the last window the last frame on the right
    first, last = next(ed.items("TS1")), list(ed.items("TS1"))[-1]
    assert all(np.array_equal(first["inputs_2d"][i], first["inputs_2d"][13]) for i in range(13))
    assert all(np.array_equal(last["inputs_2d"][i], last["inputs_2d"][13]) for i in range(14, 27))
    # the caller's arrays stay untouched (the reference centres / normalises in place); 3D is centred on joint 14
    assert test["TS1"]["data_3d"][:, 14].any() and not ed.sequence("TS1")[2][:, 14].any()
    # without the training file the scale comes from the test sequences alone (every one of them, listed or not)
    ed2 = EvalData3DHP(test, ["TS1"], 27)
    assert ed2.scale <= ed.scale and len(ed2) == 70
    assert EvalData3DHP(test, ["TS1"], 27, pos_3d_extremes=(-5.0, 3.0)).scale == 5.0
    # the npz route
    write_synth_3dhp(str(tmp_path), seed=0)
    ed3 = load_3dhp(str(tmp_path), "TS1,TS5", 27)
    assert ed3.scale == ed.scale and len(ed3) == len(ed)
    import pytest
    with pytest.raises(NotImplementedError):
        EvalData3DHP(test, ["TS1"], 27, out_all=False, stride=27)
    with pytest.raises(ValueError):
        EvalData3DHP(test, ["TS1"], 27, out_all=True, stride=9)


def test_oracle_seq2frame_windows_against_the_reference_generator():
    """oracle.gather_windows_s2f against the checksums / masks ChunkedGenerator_3dhp(out_all=False) produced (chunks_s2f.npz)."""
    from oracle import d3d_oracle as orc
    g = gold("chunks_s2f")
    kl, kr = [5, 6, 7, 11, 12, 13], [2, 3, 4, 8, 9, 10]
    for n, T in [(100, 27), (27, 27), (5, 27), (1, 9), (40, 9)]:
        rng = np.random.RandomState(n * 977 + T)
        p2 = rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32)
        p3 = rng.uniform(-1, 1, (n, 17, 3)).astype(np.float32)
        valid = (rng.uniform(0, 1, n) > 0.3).astype(np.float64)
        w, gt, m = orc.gather_windows_s2f(torch.from_numpy(p2), torch.from_numpy(p3), valid, T)
        wf, _, _ = orc.gather_windows_s2f(torch.from_numpy(p2), torch.from_numpy(p3), valid, T, True, kl, kr)
        tag = f"s2f_n{n}_T{T}"
        assert w.shape == (n, T, 17, 2) and gt.shape == (n, 1, 17, 3) and np.array_equal(gt[:, 0].numpy(), p3)
        assert np.array_equal(m.numpy(), g[tag + "/mask"])
        assert np.float64((w.numpy().astype(np.float64) * (np.arange(1, T * 34 + 1).reshape(T, 17, 2) % 97)).sum()) == g[tag + "/win_checksum"]
        assert np.float64((wf.numpy().astype(np.float64) * (np.arange(1, T * 34 + 1).reshape(T, 17, 2) % 89)).sum()) == g[tag + "/flip_checksum"]


def test_action_names_and_prefix_filter():
    """run_evaluation()'s bookkeeping (RUN:669-682, 720-734; LOAD:185-193): the action names are the first words of the loaded subjects'
    actions in first-seen order; a filter keeps the actions that START WITH the name (the reference's prefix test), in the unfiltered
    order -- so the per-action data sets partition the windows when no name is a prefix of another, and overlap when one is."""
    import numpy as np
    from diff3dhpe_amd.data import EvalData, MocapMeta
    from diff3dhpe_amd.synth import synth_mocap, SYNTH_JOINTS_LEFT as JL, SYNTH_JOINTS_RIGHT as JR
    pos, cams, kp, meta = synth_mocap(0)
    ed = EvalData(MocapMeta(pos, cams, JL, JR), kp, meta["keypoints_symmetry"], ["S9", "S11"], 27)
    assert ed.action_names() == ["Walk", "Sit", "Eat", "Wait"]
    everything = list(ed.items())
    seen = 0
    for name in ed.action_names():
        part = list(ed.items(action_filter=[name]))
        want = [it for it in everything if it["key"][1].startswith(name)]
        assert len(part) == len(want) > 0
        for a, b in zip(part, want):
            assert a["key"] == b["key"] and np.array_equal(a["inputs_2d"], b["inputs_2d"]) and np.array_equal(a["target_mask"], b["target_mask"])
        seen += len(part)
    assert seen == len(everything)
    assert len(list(ed.items(action_filter=["W"]))) == len([it for it in everything if it["key"][1][0] == "W"])      # a prefix of two names: both
    nb = sum(b["inputs_2d"].shape[0] for b in ed.batches(3, action_filter=["Walk"]))
    assert nb == len([it for it in everything if it["key"][1].startswith("Walk")])


def test_noisy_and_dropped_windows_equal_the_reference_loader():
    """The runner's robustness options (--test_extra_noise_std, --test_joint_drop; RUN:730-731, LOAD:273-290) and its per-action data sets:
    with numpy's global generator seeded as in oracle/gen_golden.py gen_dataset, EvalData.items(noise_std=, joint_drop_rate=,
    action_filter=) hands out the windows of the reference's load_Dataset(..., noise_std=, joint_drop_rate=, action_filter=) bit for bit."""
    g = gold("dataset_eval_noisy")
    ed = _data(27)
    for tag, kw in (("noise", dict(noise_std=0.02)), ("drop", dict(joint_drop_rate=0.15)),
                    ("both_walk", dict(noise_std=0.05, joint_drop_rate=0.1, action_filter=["Walk"]))):
        np.random.seed(int(g["seed"]))
        items = list(ed.items(**kw))
        for nm in ("inputs_2d", "inputs_2d_flip"):
            arr = np.stack([it[nm] for it in items])
            assert arr.dtype == np.float32 and np.array_equal(arr, g[f"{tag}/{nm}"]), (tag, nm)
    clean = np.stack([it["inputs_2d"] for it in ed.items()])
    assert (g["drop/inputs_2d"] == 0).any() and not np.array_equal(clean, g["noise/inputs_2d"])
    np.random.seed(7)
    b = next(iter(ed.batches(4, noise_std=0.02)))
    np.random.seed(7)
    assert torch.equal(b["inputs_2d"], torch.from_numpy(np.stack([it["inputs_2d"] for it in list(ed.items(noise_std=0.02))[:4]])))


def test_3dhp_noisy_and_dropped_items_equal_the_reference_loader():
    """The 3DHP runner's --test_extra_noise_std / --test_joint_drop (run_..._3dhp.py:598-600; LOAD:422-440) for both window tables and a
    per-sequence data set: EvalData3DHP.items(noise_std=, joint_drop_rate=, seq_filter=) under the seeded global numpy generator equals
    load_Dataset_3dhp(...) of the reference bit for bit (checksums + the first items of tests/golden/dataset_3dhp_eval_noisy.npz)."""
    from diff3dhpe_amd.data import EvalData3DHP
    from diff3dhpe_amd.synth import synth_mocap_3dhp
    g = gold("dataset_3dhp_eval_noisy")
    test, train = synth_mocap_3dhp(0)
    for oa, T, kw in ((False, 27, dict(noise_std=0.03, joint_drop_rate=0.1)), (True, 27, dict(noise_std=0.02)),
                      (False, 9, dict(joint_drop_rate=0.2, seq_filter="TS5"))):
        ed = EvalData3DHP(test, ["TS1", "TS5"], T, out_all=oa, train_data=train)
        np.random.seed(int(g["seed"]))
        items = list(ed.items(**kw))
        tag = f"{'s2s' if oa else 's2f'}_T{T}"
        a2, a2f = np.stack([it["inputs_2d"] for it in items]), np.stack([it["inputs_2d_flip"] for it in items])
        assert a2.dtype == np.float32 and np.array_equal(a2[:3], g[tag + "/first"])
        wts = np.arange(1, a2.size + 1, dtype=np.float64).reshape(a2.shape) % 9973.0
        assert np.float64((a2.astype(np.float64) * wts).sum()) == g[tag + "/inputs_2d_checksum"], tag
        assert np.float64((a2f.astype(np.float64) * wts).sum()) == g[tag + "/inputs_2d_flip_checksum"], tag
        assert int((a2 == 0).sum()) == int(g[tag + "/zeros"])

