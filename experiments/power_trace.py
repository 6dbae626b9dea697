#!/usr/bin/env python3
"""Socket power and shader clock of the GPU while a workload runs (on the GPU box): direct evidence for / against "the path is power-limited"
(NOTES section 000).  A sampler thread reads the amdgpu hwmon / sysfs files of every card about 20 times a second --
no GPU call, no privilege -- while the workload runs as a CHILD process (started before anything here touches the GPU; nothing is
exec'd from a process that did); falls back to `rocm-smi --json` when the sysfs files are not there.

    python experiments/power_trace.py LABEL -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
prints one JSON line: per card the power cap, and min / mean / p50 / p95 / max of power (W) and sclk (MHz) over the samples in which the
card was busy (gpu_busy_percent >= 50, or power above 1.5x the idle reading when that file is missing)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if not hw:
            continue
        h = hw[0]
        power = next((p for p in (os.path.join(h, "power1_average"), os.path.join(h, "power1_input")) if os.path.exists(p)), None)
        out.append({"dev": dev, "hwmon": h, "power": power, "cap": os.path.join(h, "power1_cap"),
                    "sclk": next((p for p in (os.path.join(h, "freq1_input"),) if os.path.exists(p)), None),
                    "dpm": os.path.join(dev, "pp_dpm_sclk"), "busy": os.path.join(dev, "gpu_busy_percent")})
    return out


def sample(cs):
    row = []
    for c in cs:
        p = _read(c["power"]) if c["power"] else None
        f = _read(c["sclk"]) if c["sclk"] else None
        if f is None:                      # pp_dpm_sclk: "0: 132Mhz\n1: 2100Mhz *"
            d = _read(c["dpm"])
            if d:
                cur = [ln for ln in d.splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    f = "".join(ch for ch in cur[0].split(":")[1] if ch.isdigit())
                    f = str(int(f) * 1000000) if f else None
        b = _read(c["busy"])
        row.append((float(p) / 1e6 if p else None, float(f) / 1e6 if f else None, float(b) if b not in (None, "") else None))
    return row


def smi_sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"], capture_output=True, text=True, timeout=5)
        d = json.loads(r.stdout)
    except Exception:
        return []
    row = []
    for k in sorted(d):
        v = d[k]
        p = next((float(v[x]) for x in v if "Power" in x and "Socket" in x or "Average Graphics Package Power" in x), None)
        f = next((float("".join(ch for ch in v[x] if ch.isdigit() or ch == ".")) for x in v if x.lower().startswith("sclk clock speed")), None)
        b = next((float(v[x]) for x in v if "GPU use" in x), None)
        row.append((p, f, b))
    return row


def stats(xs):
    xs = sorted(x for x in xs if x is not None)
    if not xs:
        return None
    q = lambda f: xs[min(len(xs) - 1, int(f * len(xs)))]
    return {"n": len(xs), "min": round(xs[0], 1), "mean": round(sum(xs) / len(xs), 1), "p50": round(q(0.5), 1), "p95": round(q(0.95), 1),
            "max": round(xs[-1], 1)}


def main():
    if "--" not in sys.argv:
        raise SystemExit(__doc__)
    i = sys.argv.index("--")
    label = sys.argv[1] if i > 1 else "run"
    cmd = sys.argv[i + 1:]
    cs = cards()
    use_smi = not cs or all(c["power"] is None for c in cs)
    idle = smi_sample() if use_smi else sample(cs)
    rows, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            rows.append((time.time(), smi_sample() if use_smi else sample(cs)))
            time.sleep(0.25 if use_smi else 0.05)
    th = threading.Thread(target=loop, daemon=True)
    t0 = time.time()
    th.start()
    child = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    stop.set()
    th.join()
    out = {"label": label, "cmd": " ".join(cmd), "seconds": round(time.time() - t0, 1), "source": "rocm-smi --json" if use_smi else "sysfs hwmon",
           "samples": len(rows), "child_rc": child.returncode, "cards": []}
    ncard = max((len(r[1]) for r in rows), default=0)
    for k in range(ncard):
        ps = [r[1][k][0] for r in rows if len(r[1]) > k]
        fs = [r[1][k][1] for r in rows if len(r[1]) > k]
        bs = [r[1][k][2] for r in rows if len(r[1]) > k]
        idle_p = idle[k][0] if len(idle) > k else None
        busy = [j for j in range(len(ps)) if (bs[j] is not None and bs[j] >= 50) or (bs[j] is None and ps[j] is not None and idle_p and ps[j] > 1.5 * idle_p)]
        cap = _read(cs[k]["cap"]) if (not use_smi and k < len(cs)) else None
        out["cards"].append({"card": k, "power_cap_W": float(cap) / 1e6 if cap else None, "idle_W": idle_p, "busy_samples": len(busy),
                             "power_W_busy": stats([ps[j] for j in busy]), "sclk_MHz_busy": stats([fs[j] for j in busy]),
                             "power_W_all": stats(ps), "sclk_MHz_all": stats(fs)})
    last = [ln for ln in (child.stdout or "").splitlines() if ln.startswith("{")]
    if last:
        try:
            d = json.loads(last[-1])
            out["workload"] = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "dtype") if k in d}
            if "machine_probes" in d:
                out["workload"]["machine_probes"] = d["machine_probes"]
        except Exception:
            pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
