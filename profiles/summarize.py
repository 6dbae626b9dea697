#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small summaries that can be committed under profiles/.

    python profiles/summarize.py stats  <rocprof_dir> <out.md>        # --kernel-trace --stats run
    python profiles/summarize.py pmc    <fetch_dir> <write_dir> <out.json> [label]   # two --pmc runs (FETCH_SIZE / WRITE_SIZE)

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes (TCC slots), both are in KiB-units of 1024 B... (FETCH_SIZE = TCC_EA0_RDREQ x 64 B reported /1024), and on gfx950
FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x, so read bytes are doubled before use.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


_DEMANGLE = {}


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLE:
        out = name
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                import subprocess
                out = subprocess.run([tool, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
                break
            except Exception:
                continue
        _DEMANGLE[name] = out
    return _DEMANGLE[name]


def short(name: str) -> str:
    name = demangle(name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("d3d::", "")
    return name.strip()[:70]


def find(dirpath, pattern):
    hits = glob.glob(os.path.join(dirpath, "**", pattern), recursive=True)
    return sorted(hits)


def stats(dirpath, out):
    files = find(dirpath, "*kernel_stats.csv")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    with open(out, "w") as fh:
        fh.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            fh.write("| {} | {} | {:.3f} | {:.1f} | {:.1f} | {:.1f} | {:.2f} |\n".format(
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print(open(out).read())


def pmc_table(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in find(dirpath, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                a = acc[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def pmc(fetch_dir, write_dir, out, label=""):
    fe, wr = pmc_table(fetch_dir, "FETCH_SIZE"), pmc_table(write_dir, "WRITE_SIZE")
    res = {"label": label, "note": "per-launch means; fetch_bytes = 2 * FETCH_SIZE * 1024 (gfx950 half-count correction), "
                                    "write_bytes = WRITE_SIZE * 1024", "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        res["kernels"][k] = {"launches": max(nf, nw), "FETCH_SIZE_raw_KiB": round(f, 1), "WRITE_SIZE_raw_KiB": round(w, 1),
                             "fetch_bytes": round(2 * f * 1024), "write_bytes": round(w * 1024),
                             "hbm_bytes": round(2 * f * 1024 + w * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
