"""Build libd3d_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python -m diff3dhpe_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libd3d_hip.so")
SOURCES = ["engine.hip", "kernels_gemm.hip", "kernels_gemm_f16x3.hip", "kernels_gemm_x3p.hip", "kernels_elem.hip", "kernels_attn.hip", "kernels_attn_x3.hip", "kernels_attn_bf16.hip", "kernels_qkv_sattn.hip", "kernels_qkv_tattn.hip", "kernels_fc1_x3.hip", "kernels_proj_x3.hip", "kernels_fc2_ring.hip", "kernels_gemm_bf16q.hip", "probes.hip"]
HEADERS = ["d3d_kernels.h", "gemm_x3p_prelude.h", "gemm_x3p_epilogue.h", "x3q_epilogue_acc.h", "qkv_fused_kloop.h", os.path.join("..", "..", "include", "d3d.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The row kernels are plain scalar code; the SLP vectoriser packs it into v_pk_*_f32 with op_sel broadcasts, one form of which deviates
# on gfx950 beside another wave's MFMAs (x3q_epilogue_acc.h splat2_rt; tests/test_abi_host.py pins its absence).  They are memory-bound:
# nothing to gain from the packing.
EXTRA_FLAGS = {"kernels_elem.hip": ["-fno-slp-vectorize"]}


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
