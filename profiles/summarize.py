#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small summaries that can be committed under profiles/.

    python profiles/summarize.py stats  <rocprof_dir> <out.md>        # --kernel-trace --stats run
    python profiles/summarize.py pmc    <fetch_dir> <write_dir> <out.json> [label]   # two --pmc runs (FETCH_SIZE / WRITE_SIZE)
    python profiles/summarize.py sq     <sq_dir> <out.json>           # one --pmc run of SQ / GRBM counters
    python profiles/summarize.py traffic <pmc.json> <hbm_traffic.json> <T> <B> <precision>   # per-class bytes for bench.py

profiles/collect.sh runs all the passes on the GPU box and calls these.

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
        if out.startswith("_Z"):   # c++filt without _Float16 (DF16_) support: recover "name<ints>" from the mangled form
            m = re.search(r"\d+(k_[A-Za-z0-9_]+?)I((?:Li\d+E)+)E", out)
            if m:
                out = "{}<{}>".format(m.group(1), ", ".join(re.findall(r"Li(\d+)E", m.group(2))))
        _DEMANGLE[name] = out
    return _DEMANGLE[name]


_EPI = {"0": "EPI_NONE", "1": "EPI_GELU", "2": "EPI_RESIDUAL"}
_OUT = {"0": "fp32-out", "1": "planes-out", "2": "pair-out", "3": "bf16-out"}


def short(name: str) -> str:
    name = demangle(name).replace("(anonymous namespace)::", "")
    if "k_qkv_sattn" in name:
        return "k_qkv_sattn<255 rows x 192 cols persistent, LN-folded qkv GEMM + 17-key attention from LDS> (spatial blocks)"
    if "k_qkv_tattn<true>" in name or "k_qkv_tattnILb1E" in name:
        return "k_qkv_tattn<256 rows (the frames of 255 / T joints of a batch element) x 192 cols persistent, LN-folded qkv GEMM + per-joint T-key attention from LDS> (temporal blocks)"
    if "k_qkv_tattn" in name:
        return "k_qkv_tattn<256 rows (the frames of one joint) x 192 cols persistent, LN-folded qkv GEMM + T-key attention from LDS> (temporal blocks)"
    if "k_fc1_x3" in name:
        return "k_fc1_x3<256x256 persistent, EPI_GELU, pair-out, LN-folded> (fc1)"
    if "k_proj_x3" in name:
        return "k_proj_x3<192x256 persistent, EPI_RESIDUAL, pair-out, plane residual + row stats> (proj)"
    if "k_gemm_bf16q" in name:      # template argument = EPI: 0 none (qkv), 1 GELU (fc1)
        gelu = "k_gemm_bf16q<1>" in name or "k_gemm_bf16qILi1E" in name
        return "k_gemm_bf16q<256x256 persistent, {}, bf16-out, bf16 operands> ({})".format("EPI_GELU" if gelu else "EPI_NONE", "fc1" if gelu else "qkv")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("d3d::", "").strip()
    _FX = {"0": "", "1": ", LN-folded", "2": ", plane residual", "6": ", plane residual + row stats",
           "10": ", plane residual + post-norm in the epilogue", "16": ", bf16 operands",
           "24": ", bf16 operands, whole rows + LayerNorm(s) in the epilogue"}
    m = re.match(r"k_linear_x3q(_persist)?<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:   # <TM, WM, WN, EPI, OUTSPLIT, FX> -> tile and role
        per, tm, wm, wn, epi, osp, fx = m.groups()
        tile = "{}x{}{}".format(16 * int(tm) * int(wm), 64 * int(wn), " persistent" if per else "")
        role = {("0", "0"): "qkv", ("0", "1"): "qkv", ("2", "0"): "proj, fc2" if (fx or "0") == "0" else "fc2", ("2", "2"): "proj",
                ("1", "2"): "fc1"}.get((epi, osp), "")
        if fx == "10":
            role = "fc2 + post-norm" + (", last block" if osp == "0" else "")
        if fx == "16":
            role = {("0", "3"): "qkv", ("1", "3"): "fc1", ("2", "0"): "proj / fc2"}.get((epi, osp), "")
        if fx == "24":
            role = "proj + norm2 / fc2 + post-norm + next norm1"
        return "k_linear_x3q<{}, {}, {}{}>{}".format(tile, _EPI[epi], _OUT[osp], _FX.get(fx or "0", ", fx" + str(fx)),
                                                   " (" + role + ")" if role else "")
    if "k_qkv_sattn" in name:
        return "k_qkv_sattn<255 rows x 192 cols persistent, LN-folded qkv GEMM + 17-key attention from LDS> (spatial blocks)"
    m = re.match(r"k_attn_temporal_x3<(\d+), (\d+)>", name)
    if m:
        return "k_attn_temporal_x3<{} key tiles, {} units/wg>{}".format(m.group(1), m.group(2), " (spatial blocks)" if m.group(2) != "1" else " (temporal blocks)")
    m = re.match(r"k_attn_temporal_x3p<1, (\d+), (\d+)>", name)
    if m:   # wave-private persistent form: 8 units per workgroup = spatial blocks (17 joints), 6 = temporal blocks of T <= 32
        return "k_attn_temporal_x3p<1, {}, {}>{}".format(m.group(1), m.group(2), " (spatial blocks)" if m.group(1) == "8" else " (temporal blocks)")
    m = re.match(r"k_attn_bf16<(\d+), (\d+), (\d+)>", name)
    if m:
        return "k_attn_bf16<{} key tiles, {} units/wg, {} query tiles/wave>{}".format(m.group(1), m.group(2), m.group(3), " (spatial blocks)" if m.group(2) != "1" else " (temporal blocks)")
    return name[:70]


def find(dirpath, pattern):
    """Files of ONE profiled process: rocprofv3 names its outputs <pid>_*.csv, and gpurun merges the directories of
    successive calls, so an older run's files can sit next to the newest -- keep the newest pid's only."""
    hits = glob.glob(os.path.join(dirpath, "**", pattern), recursive=True)
    if not hits:
        return []
    newest = max(hits, key=os.path.getmtime)
    pid = os.path.basename(newest).split("_")[0]
    return sorted(h for h in hits if os.path.basename(h).split("_")[0] == pid)


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


def sq(dirpath, out):
    """Sums of every collected counter per kernel over all its launches (SQ_* are in quad-cycles except
    SQ_VALU_MFMA_BUSY_CYCLES; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
    acc = defaultdict(lambda: defaultdict(float))
    for f in find(dirpath, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {k: {c: round(v) for c, v in sorted(d.items())} for k, d in acc.items()}
    for k, d in res.items():
        if d.get("SQ_BUSY_CU_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # MFMA pipe busy share of the time a CU had waves: MFMA_BUSY sums the 4 SIMDs of a CU, in the unit of BUSY_CU_CYCLES
            d["mfma_busy_fraction"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"]), 4)
        if d.get("SQ_WAVE_CYCLES"):
            d["wave_time_parked_fraction"] = round(d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"], 4)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:5000])


def traffic(pmc_json, traffic_json, T, B, prec):
    """Mean HBM-side bytes per launch of each bench.py kernel class -> profiles/hbm_traffic.json (read by bench.py)."""
    k = json.load(open(pmc_json))["kernels"]
    # kernel name patterns per class; the fp16-MFMA attention serves both block types from one template: <1, MU> = groups of
    # <= 32 tokens = the spatial blocks, everything else (x3s<8>, x3p<3, 1>, x3p<8, 1>, x3<NKT, 1>) the temporal blocks
    def is_spatial(n):
        if "(temporal blocks)" in n:
            return False
        if "k_qkv_sattn" in n or "k_qkv_tattn" in n:
            return False
        return "k_attn_spatial" in n or "k_attn_temporal_x3p<1," in n or "k_attn_temporal_x3<1," in n or "(spatial blocks)" in n
    cls = {"linear": lambda n: "k_linear" in n or "k_fc1_x3" in n or "k_proj_x3" in n or "k_gemm_bf16q" in n, "layernorm": lambda n: "k_layernorm" in n, "qkv_sattn": lambda n: "k_qkv_sattn" in n, "qkv_tattn": lambda n: "k_qkv_tattn" in n,
           "attn_spatial": is_spatial, "attn_temporal": lambda n: ("k_attn_temporal" in n or "k_attn_bf16" in n) and not is_spatial(n)}
    tj = json.load(open(traffic_json)) if os.path.exists(traffic_json) else {}
    for c, pat in cls.items():
        sel = [v for n, v in k.items() if pat(n)]
        n = sum(v["launches"] for v in sel)
        if n:
            tj["{}:T{}:B{}:{}".format(c, T, B, prec)] = round(sum(v["hbm_bytes"] * v["launches"] for v in sel) / n)
    json.dump(tj, open(traffic_json, "w"), indent=1)
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:7])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
