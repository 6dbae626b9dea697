"""The plain-C oracle (oracle/ddim_index_oracle.c: integer schedule + scalar DDIM update) against the golden vectors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import gold, ROOT
from helpers import inputs
from oracle import d3d_oracle as orc


@pytest.fixture(scope="module")
def clib(tmp_path_factory):
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_c.so")
    src = os.path.join(ROOT, "oracle", "ddim_index_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"], check=True)
    lib = C.CDLL(so)
    lib.oracle_ddim_update.restype = C.c_float
    lib.oracle_ddim_update.argtypes = [C.c_float] * 7
    return lib


def test_c_schedule_every_S(clib):
    g = gold("ddim_times_N1000")
    flat, offs = g["flat"], g["offsets"]
    for S in range(1, 1001):
        out = (C.c_int32 * (S + 1))()
        assert clib.oracle_ddim_times(1000, S, out) == 0
        assert list(out) == flat[offs[S - 1]:offs[S]].tolist(), S


def test_c_ddim_update_reproduces_reference_trajectory(clib):
    g = gold("ddim_small_T81_S5")
    tabs = orc.diffusion_tables("cosine", 1000)
    ac, so = tabs["alphas_cumprod"].numpy(), tabs["sqrt_one_minus_alphas_cumprod"].numpy()
    times = orc.ddim_times(1000, 5)
    y = inputs(int(g["B"]), 81, int(g["input_seed"]))["noise"].numpy().reshape(-1)
    x0s, rev = g["x_start_est"], g["x_reverse_diffusion"]
    idx = np.random.RandomState(0).choice(y.size, 300, replace=False)
    for step in range(4):
        t, tn = times[step], times[step + 1]
        x0 = x0s[..., step].reshape(-1)
        want = rev[..., step].reshape(-1)
        for i in idx:
            got = clib.oracle_ddim_update(float(x0[i]), float(y[i]), 0.0, float(ac[t]), float(ac[tn]), float(so[t]), 0.0)
            assert got == want[i], (step, i, got, want[i])          # bit-exact
        y = want
