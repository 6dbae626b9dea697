"""Host logic without a GPU: the import shim that lets the unchanged reference runner pick up the engine, and the
one-process-per-GPU sharding / all-gather / reduction logic under a world-size-2 gloo group."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import ROOT
from diff3dhpe_amd import parallel


def test_compat_shim_resolves_engine_and_reference_modules(tmp_path):
    # a stand-in "reference checkout": namespace package common/ with some other module in it
    (tmp_path / "common").mkdir()
    (tmp_path / "common" / "loss.py").write_text("MARK = 'reference-side module'\n")
    (tmp_path / "common" / "nets").mkdir()
    (tmp_path / "common" / "nets" / "load_net.py").write_text("raise RuntimeError('reference load_net must be shadowed')\n")
    code = textwrap.dedent("""
        import sys
        from common.nets.load_net import HPE_model
        from common.conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames import GaussianDiffusion
        from common.conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames import GaussianDiffusion as G2
        import common.loss
        import diff3dhpe_amd
        assert HPE_model is diff3dhpe_amd.HPE_model and GaussianDiffusion is diff3dhpe_amd.GaussianDiffusion is G2
        assert common.loss.MARK == 'reference-side module'
        print('shim ok')
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "diff3dhpe_amd", "compat")]))
    out = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "shim ok" in out.stdout, out.stderr


def test_shard_bounds_partition():
    for B in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["D3D_ROOT"])
    from diff3dhpe_amd import parallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    for B in (8, 7, 1):                      # even, ragged, fewer rows than ranks
        full = torch.randn(B, 5, 17, 3)
        gt = torch.randn(B, 5, 17, 3)
        lo, hi = parallel.shard_bounds(B, rank, world)
        local = full[lo:hi] * 2.0            # stand-in for the per-rank sampling result
        got = parallel.all_gather_pred(local, B)
        assert got.shape == full.shape and torch.equal(got, full * 2.0), (B, rank)
        # the flip-TTA pair in ONE collective (evaluate.py): (b, 2, T, J, 3) shards
        both = parallel.all_gather_pred(torch.stack([local, -local], dim=1), B)
        assert both.shape == (B, 2, 5, 17, 3) and torch.equal(both[:, 0], full * 2.0) and torch.equal(both[:, 1], full * -2.0), (B, rank)
        # frame-weighted MPJPE reduction: per-rank partial sums == global sums
        err = (local - gt[lo:hi]).norm(dim=-1)
        s, c = parallel.reduce_sums(float(err.sum()), err.numel(), torch.device("cpu"))
        ref = (full * 2.0 - gt).norm(dim=-1)
        assert abs(s - float(ref.sum())) < 1e-3 and c == ref.numel(), (B, rank, s, c)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_allgather_and_reduction_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, D3D_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, e[-2000:]


def _bench(args, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, cwd=ROOT,
                          timeout=timeout)


def test_bench_self_launch_starts_one_rank_process_per_gpu():
    """`python bench.py --gpus N` with no launcher: the parent (which never imports torch) starts N rank processes with the
    torch.distributed rendezvous environment, relays rank 0's stdout alone and returns 0 (launch-check mode: no GPU needed)."""
    import json
    run = _bench(["--gpus", "4"], {"D3D_BENCH_LAUNCH_CHECK": "1"})
    assert run.returncode == 0, run.stderr
    out = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world_size"] == 4 and out[0]["launcher"] == "self"
    err = [json.loads(l) for l in run.stderr.splitlines() if l.startswith("{")]
    assert sorted(e["rank"] for e in err) == [1, 2, 3] and all(e["local_rank"] == e["rank"] for e in err)
    assert {e["master_port"] for e in err} == {out[0]["master_port"]} and out[0]["master_addr"] == "127.0.0.1"


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies must not leave the others waiting in a collective: the launcher ends its own children and returns the
    failing rank's exit code (here rank 2 exits 7 while the others would sleep 30 s)."""
    import time
    t0 = time.time()
    run = _bench(["--gpus", "3"], {"D3D_BENCH_LAUNCH_CHECK": "1", "D3D_BENCH_LAUNCH_CHECK_FAIL": "2"})
    assert run.returncode == 7 and "rank 2 failed" in run.stderr, (run.returncode, run.stderr)
    assert time.time() - t0 < 20


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    run = _bench(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "D3D_BENCH_LAUNCH_CHECK": ""})
    assert run.returncode != 0 and "WORLD_SIZE=3" in (run.stderr + run.stdout)


def test_bench_self_launch_eight_rank_processes():
    """The N = 8 launch the driver makes at round end, without a GPU (launch-check mode): eight rank processes, ONE JSON line on stdout."""
    import json
    run = _bench(["--gpus", "8"], {"D3D_BENCH_LAUNCH_CHECK": "1"})
    assert run.returncode == 0, run.stderr
    out = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
    err = [json.loads(l) for l in run.stderr.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["world_size"] == 8 and sorted(e["rank"] for e in err) == list(range(1, 8))


def test_bench_self_launch_ends_its_ranks_when_the_launcher_is_terminated():
    """a harness that ends only the launcher (SIGTERM) must not orphan rank processes that wait in a collective -- the
    launcher terminates its children (their own sessions), waits, kills, and exits non-zero."""
    import signal
    import time
    import psutil
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"D3D_BENCH_LAUNCH_CHECK": "1", "D3D_BENCH_LAUNCH_CHECK_FAIL": "99"})     # no rank is 99: every rank sleeps 30 s
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    kids = []
    t0 = time.time()
    while len(kids) < 3 and time.time() - t0 < 20:
        kids = psutil.Process(p.pid).children()
        time.sleep(0.1)
    assert len(kids) == 3
    time.sleep(0.5)
    p.send_signal(signal.SIGTERM)
    _, err = p.communicate(timeout=20)
    assert p.returncode not in (0, None) and "ending the rank processes" in err
    gone, alive = psutil.wait_procs(kids, timeout=5)
    assert not alive


def test_bench_host_side_helpers():
    """bench.py pieces that run without a GPU: the per-row synthetic batch (any shard of it holds the same rows), the cgroup-aware CPU
    count, and the concurrent CPU-baseline leg (oracle workers as fresh processes; one process when the CPU quota holds only one)."""
    import importlib.util
    import numpy as np
    from diff3dhpe_amd.synth import synth_inputs_rows
    a = synth_inputs_rows(0, 7, 9)
    b = synth_inputs_rows(3, 5, 9)
    assert a["x2d"].shape == (7, 9, 17, 2) and all(np.array_equal(a[k][3:5], b[k]) for k in a)
    assert synth_inputs_rows(4, 4, 9)["noise"].shape == (0, 9, 17, 3)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    aff, quota = bench.usable_cpus()
    assert aff >= 1 and (quota is None or quota > 0)
    one = bench.cpu_baseline_concurrent(9, 3, 0, 4, 64, 0.1, quota=4.0)          # a 4-CPU quota holds ONE 4-thread process: nothing is started
    assert one["equal_to_single_process"] and one["processes"] == 1 and one["value"] is None
    two = bench.cpu_baseline_concurrent(9, 3, 0, 1, 2, 0.1, run_s=0.5, quota=None)   # two 1-thread oracle workers, T = 9
    assert two.get("processes") == 2 and two["value"] and two["steps_finished"] >= 2, two


def test_bench_power_sampler_reads_the_card_with_this_pci_address(tmp_path):
    """bench.py's PowerSampler (the `power` object of the bench line: socket power and shader clock of the rank's card over the timed region,
    from the amdgpu hwmon files): it picks the card whose sysfs device resolves to the rank's PCI address, reports microwatts / hertz as W /
    MHz with the cap beside them, and is silent (None) when no card matches or the tree is missing."""
    import importlib.util
    import time as _t
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n, addr, pw in ((0, "0000:05:00.0", 240e6), (1, "0000:15:00.0", 1394e6)):
        real = tmp_path / "pci" / addr
        hw = real / "hwmon" / f"hwmon{n + 3}"
        hw.mkdir(parents=True)
        (hw / "power1_input").write_text(str(int(pw)))
        (hw / "freq1_input").write_text(str(int(1878e6)))
        (hw / "power1_cap").write_text(str(int(1400e6)))
        (tmp_path / "drm" / f"card{n}").mkdir(parents=True)
        os.symlink(real, tmp_path / "drm" / f"card{n}" / "device")
    ps = bench.PowerSampler("0000:15:00.0", root=str(tmp_path / "drm"))
    assert ps.files and ps.files["power"].endswith("hwmon4/power1_input")
    ps.start()
    _t.sleep(0.2)
    out = ps.stop()
    assert out["socket_power_W"]["p50"] == 1394.0 and out["power_cap_W"] == 1400.0 and out["sclk_MHz"]["p50"] == 1878.0
    assert out["frac_of_cap_p50"] == round(1394 / 1400, 4) and out["samples"] >= 2
    none = bench.PowerSampler("0000:99:00.0", root=str(tmp_path / "drm"))
    none.start()
    assert none.files is None and none.stop() is None
    assert bench.PowerSampler("0000:15:00.0", root=str(tmp_path / "nowhere")).stop() is None


def test_bench_self_launch_watchdog_names_a_rank_that_never_finishes_its_first_sampling():
    """a rank that has not finished a first sampling `D3D_BENCH_STARTUP_TIMEOUT_S` after the start (a hung
    rendezvous, a device that does not come up) is NAMED, every rank is ended and the launcher exits non-zero -- it does not sit until the
    harness's own limit.  Launch-check mode: rank 1 sleeps before it reports ready, the others report and then wait."""
    import time
    t0 = time.time()
    run = _bench(["--gpus", "3"], {"D3D_BENCH_LAUNCH_CHECK": "1", "D3D_BENCH_LAUNCH_CHECK_HANG": "1", "D3D_BENCH_STARTUP_TIMEOUT_S": "3"})
    assert run.returncode == 124, (run.returncode, run.stderr)
    assert "rank(s) [1] had not finished a first sampling" in run.stderr and "ending all ranks" in run.stderr
    assert time.time() - t0 < 25


def test_bench_self_launch_watchdog_is_quiet_when_every_rank_reports():
    """All ranks report their first sampling: the watchdog never fires even with a tiny limit once they have all reported."""
    run = _bench(["--gpus", "2"], {"D3D_BENCH_LAUNCH_CHECK": "1", "D3D_BENCH_STARTUP_TIMEOUT_S": "30"})
    assert run.returncode == 0 and "had not finished" not in run.stderr, run.stderr
