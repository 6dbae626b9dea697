#!/usr/bin/env python3
"""Cross-timing of the CPU oracle against the imported reference -- TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.

bench.py's `cpu_baseline` times oracle/d3d_oracle.py (kind "port") on the GPU node, where the reference cannot travel.  This
script shows, where the reference IS importable, that the port costs what the reference costs: the same DDIM sampling
(T, S, B, random-init D=512 depth 8, same threads) through both, best of `--reps`, outputs compared.
Writes profiles/r02_cpu_oracle_vs_reference.json.      python oracle/crosstime_reference.py [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle.gen_golden as gg          # noqa: E402  (imports the reference with the DropPath stub)
from oracle import d3d_oracle as orc    # noqa: E402
from diff3dhpe_amd.spec import DenoiserConfig  # noqa: E402
from diff3dhpe_amd.synth import synth_inputs   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    rows = []
    for T, S, B in ((243, 9, 1), (81, 9, 1), (81, 9, 4)):
        cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
        net, diff, sd = gg.build_ref(cfg, 0, sampling=S)
        inp = synth_inputs(B, T, seed=42)
        x2d, nz = torch.from_numpy(inp["x2d"]), torch.from_numpy(inp["noise"])
        tabs = orc.diffusion_tables("cosine", 1000)
        t_ref, t_orc = [], []
        for _ in range(a.reps):
            t0 = time.time()
            with gg.inject_noise(nz), torch.no_grad():
                y_ref = diff.ddim_sample_loop(x2d, list(nz.shape))
            t_ref.append(time.time() - t0)
            t0 = time.time()
            y_orc = orc.ddim_sample_loop(sd, tabs, x2d, nz, num_timesteps=1000, sampling_timesteps=S, depth=8)
            t_orc.append(time.time() - t0)
        rows.append({"T": T, "S": S, "B": B, "reference_s": round(min(t_ref), 3), "oracle_s": round(min(t_orc), 3),
                     "oracle_over_reference": round(min(t_orc) / min(t_ref), 3), "max_abs_diff": float((y_ref - y_orc).abs().max()),
                     "reference_seq_per_s": round(B / min(t_ref), 4), "oracle_seq_per_s": round(B / min(t_orc), 4)})
        print(rows[-1], flush=True)
    out = {"host": {"cpus": threads, "torch": torch.__version__}, "what": "min of %d runs each; same inputs, weights, noise, thread count" % a.reps,
           "cases": rows}
    path = os.path.join(os.path.dirname(HERE), "profiles", "r02_cpu_oracle_vs_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
