#!/usr/bin/env python3
"""Bisecting harness for the two-processes-on-one-GPU nondeterminism.

    python experiments/bisect_two_proc.py [--procs 2] [--rounds 10] [--iters 6] [--frames 27] [--batch 3] [--sampling 3] [env...]

The parent never touches the GPU.  Each round it starts `--procs` children AT ONCE; every child builds the same engine,
turns the library's debug trace on (d3d_engine_set_trace: a checksum launch after every kernel of the block flow over the
rows it has just written) and runs the same DDIM sampling `--iters` times.  Every trace of every child of every round must be
identical; the parent takes the per-entry majority as the reference and prints, for every deviating run, the FIRST entry that
differs (forward, block, kernel, buffer) and how many differ after it -- i.e. the kernel whose output moved first.
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KN = {0: "embed", 1: "entry", 2: "qkv", 3: "attn", 4: "proj", 5: "fc1", 6: "fc2pn", 7: "rowk", 8: "head", 9: "head-again"}


def child(a):
    sys.path.insert(0, ROOT)
    import torch
    import diff3dhpe_amd as d3d
    from diff3dhpe_amd.spec import DenoiserConfig
    from diff3dhpe_amd.synth import synth_state_dict, synth_inputs
    T, B, S = a.frames, a.batch, a.sampling
    dev = torch.device("cuda:0")
    cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=8)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=8)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 0).items()})
    net.precision = "f16x3"
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=S, loss_type="l2", clip_denoised=True,
                                 beta_schedule="cosine", ddim_sampling_eta=0.0).eval().to(dev)
    eng = diff._engine(dev)
    inp = synth_inputs(B, T, seed=42)
    x2d = torch.from_numpy(inp["x2d"]).to(dev)
    noise = torch.from_numpy(inp["noise"]).to(dev)
    cap = (S * (16 * 10 + 8) + 16) * a.views
    if not a.no_trace:
        eng.set_trace(cap, a.views)
    sums, outs, tags = [], [], None
    for _ in range(a.iters):
        y = eng.ddim_sample(x2d, noise)
        if not a.no_trace:
            tr = eng.trace_read()
            tags = [t for t, _ in tr]
            sums.append([s for _, s in tr])
        outs.append(y.cpu().numpy().view(np.uint32).astype(np.uint64).sum())
        if a.recheck:   # the last forward's head once more, twice, from the stream the workspace still holds (w.X = offset 0)
            M = B * T * 17
            X = eng._ws[: M * 512 * 4].view(torch.float32).view(M, 512)
            ya = eng.head(X).clamp(-1.0, 1.0)
            yb = eng.head(X).clamp(-1.0, 1.0)
            y0 = y.reshape(M, 3)
            for nm, u, v in (("sampler-vs-again1", y0, ya), ("again1-vs-again2", ya, yb)):
                d = (u != v)
                if d.any():
                    rows = d.any(1).nonzero().flatten().tolist()
                    i = rows[0]
                    print(f"RECHECK {nm}: {int(d.sum())} values in {len(rows)} rows differ; rows {rows[:16]}{'...' if len(rows) > 16 else ''}; "
                          f"row {i}: {u[i].tolist()} vs {v[i].tolist()}; X row finite {bool(torch.isfinite(X[i]).all())}", flush=True)
                    # which part of the dot product is off?  per-lane contributions (lane l owns columns 4l..4l+3 and 256+4l..+3)
                    sd = net.state_dict()
                    g, bb = sd["head.0.weight"].double().to(dev), sd["head.0.bias"].double().to(dev)
                    W, hb = sd["head.1.weight"].double().to(dev), sd["head.1.bias"].double().to(dev)
                    for i in rows[:3]:
                        x = X[i].double()
                        xn = (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-5) * g + bb
                        k = int(d[i].nonzero()[0])
                        pr = (xn * W[k]).view(2, 64, 4).sum((0, 2))          # per-lane partial of o[k]
                        exact = float(pr.sum() + hb[k])
                        P = (xn.view(1, 2, 64, 4) * W.view(3, 2, 64, 4))            # [k', i, lane, e] element products
                        pl = P.sum(3)                                                # [k', i, lane] float4 partials
                        for nm2, val in (("first", float(u[i, k])), ("second", float(v[i, k]))):
                            err = val - max(-1.0, min(1.0, exact))
                            if abs(err) < 1e-5 or abs(val) >= 1.0:
                                continue
                            hyp = {}
                            for ii in range(2):
                                hyp[f"half {ii} dropped"] = -float(pl[k, ii].sum())
                                for kk in range(3):
                                    if kk != k:
                                        hyp[f"half {ii} of row k'={kk} instead"] = float(pl[kk, ii].sum() - pl[k, ii].sum())
                                hyp[f"half {ii}: other half's weights"] = float((xn.view(2, 64, 4)[ii] * W[k].view(2, 64, 4)[1 - ii]).sum() - pl[k, ii].sum())
                            best = min(hyp, key=lambda h: abs(hyp[h] - err))
                            cands = []
                            for ii in range(2):
                                for sign, nm3 in ((-1.0, "dropped"), (1.0, "doubled")):
                                    dd = (sign * pl[k, ii] - err).abs()
                                    cands.append((float(dd.min()), f"lane {int(dd.argmin())} float4 {ii} {nm3}"))
                                    de = (sign * P[k, ii] - err).abs()
                                    cands.append((float(de.min()), f"element lane {int(de.argmin()) // 4} i {ii} e {int(de.argmin()) % 4} {nm3}"))
                                for kk in range(3):
                                    if kk != k:
                                        dd = ((pl[kk, ii] - pl[k, ii]) - err).abs()
                                        cands.append((float(dd.min()), f"lane {int(dd.argmin())} float4 {ii} took row k'={kk}"))
                            # a contiguous block of lanes (multiples of 4) whose half-ii partial is missing / doubled: what a DPP or
                            # LDS read of the accumulator register would see if the last packed add had not yet landed in those lanes
                            for ii in range(2):
                                cs = torch.cat([torch.zeros(1, dtype=pl.dtype, device=pl.device), pl[k, ii].cumsum(0)])
                                for a0 in range(0, 64, 4):
                                    for b0 in range(a0 + 4, 65, 4):
                                        blk = float(cs[b0] - cs[a0])
                                        cands.append((abs(-blk - err), f"lanes [{a0},{b0}) half {ii} missing"))
                            cands.sort()
                            print(f"   row {i} k {k} err {err:+.6e} | wave-level best: {best} ({hyp[best]:+.6e}) | lane-level best: {cands[0][1]} (residual {cands[0][0]:.2e}), "
                                  f"next {cands[1][1]} ({cands[1][0]:.2e})", flush=True)
    np.savez(a.out, sums=np.array(sums, dtype=np.uint64), tags=np.array(tags or [], dtype=np.uint32),
             outs=np.array(outs, dtype=np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--frames", type=int, default=27)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--sampling", type=int, default=3)
    ap.add_argument("--views", type=int, default=1, help="1..8: sum every buffer through this many XCD 'views' (they must agree)")
    ap.add_argument("--no-trace", action="store_true", help="outputs only (control: does the trace itself hide the effect?)")
    ap.add_argument("--recheck", action="store_true", help="after every sampling, run the head twice more on the final stream and print what differs")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.child:
        return child(a)
    tmp = tempfile.mkdtemp(prefix="bisect_")
    runs = []   # (round, proc, iter, sums row, out)
    tags = None
    for r in range(a.rounds):
        files = [os.path.join(tmp, f"r{r}_p{p}.npz") for p in range(a.procs)]
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "--out", f, "--iters", str(a.iters),
                                "--frames", str(a.frames), "--batch", str(a.batch), "--sampling", str(a.sampling), "--views", str(a.views)]
                               + (["--no-trace"] if a.no_trace else []) + (["--recheck"] if a.recheck else []), cwd=ROOT) for f in files]
        rcs = [p.wait() for p in ps]
        if any(rcs):
            print("child failed", rcs, flush=True)
            sys.exit(1)
        for p, f in enumerate(files):
            d = np.load(f)
            if d["tags"].size:
                tags = d["tags"]
            for i in range(a.iters):
                runs.append((r, p, i, d["sums"][i] if d["sums"].size else None, int(d["outs"][i])))
        print(f"round {r} done", flush=True)
    outs = np.array([x[4] for x in runs], dtype=np.uint64)
    vals, cnts = np.unique(outs, return_counts=True)
    ref_out = vals[np.argmax(cnts)]
    nbad = int((outs != ref_out).sum())
    print(f"{len(runs)} samplings ({a.procs} procs x {a.rounds} rounds x {a.iters} iters): {nbad} with a deviating final output")
    if a.no_trace or tags is None:
        for (r, p, i, _, o) in runs:
            if o != ref_out:
                print(f"  round {r} proc {p} iter {i}")
        return
    S = np.stack([x[3] for x in runs])
    if a.views > 1:   # self-consistency: the views of one buffer (consecutive entries, same tag below bit 28) must agree
        base = tags & 0x0FFFFFFF
        starts = [j for j in range(len(tags)) if (tags[j] >> 28) == 0]
        nbadv = 0
        for k, (r, p, i, row, o) in enumerate(runs):
            for j in starts:
                grp = row[j:j + a.views]
                if (grp != grp[0]).any():
                    t = int(base[j])
                    vals, cnt = np.unique(grp, return_counts=True)
                    odd = [int(v) for v in range(a.views) if grp[v] != vals[np.argmax(cnt)]]
                    print(f"  VIEWS DISAGREE round {r} proc {p} iter {i}: forward {(t >> 16) & 0xFFF} block {(t >> 8) & 255} kernel "
                          f"{KN.get((t >> 4) & 15, '?')} buffer {t & 15}: deviating views {odd}")
                    nbadv += 1
        print(f"buffers whose XCD views disagree: {nbadv}")
    ref = np.empty(S.shape[1], dtype=np.uint64)
    for j in range(S.shape[1]):
        v, c = np.unique(S[:, j], return_counts=True)
        ref[j] = v[np.argmax(c)]
    first_hist = {}
    for k, (r, p, i, row, o) in enumerate(runs):
        bad = np.nonzero(row != ref)[0]
        if bad.size == 0:
            continue
        t = int(tags[bad[0]])
        key = (KN.get((t >> 4) & 15, "?"), t & 15)
        first_hist[key] = first_hist.get(key, 0) + 1
        print(f"  round {r} proc {p} iter {i}: first deviation at entry {bad[0]} = view {t >> 28} forward {(t >> 16) & 0xFFF} block {(t >> 8) & 255} "
              f"kernel {KN.get((t >> 4) & 15, '?')} buffer {t & 15}; {bad.size} of {S.shape[1]} entries differ; "
              f"final output {'differs' if o != ref_out else 'same'}")
    print("first-deviation histogram (kernel, buffer):", first_hist)


if __name__ == "__main__":
    main()
