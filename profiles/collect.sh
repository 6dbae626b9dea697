#!/bin/bash
# Run ON THE GPU BOX (through gpurun, from the repo root): every rocprofv3 pass of the default bench workload, each in
# its own run (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ/GRBM counters -- counters never combined with traces),
# condensed into profiles/<tag>_*.  Raw output stays under gpurun_out/ (scratch).
#   profiles/collect.sh r03_f16x3 [bench.py args...]          e.g.  profiles/collect.sh r03_T81 --frames 81 --batch 128
# The profiled bench runs ONE stream, without its own event-timed pass (--streams 1 --profile-steps 0): kernel durations in the
# trace are then those of kernels running alone, which is what the per-kernel roofline figures mean.
set -eo pipefail
tag=${1:-r04_f16x3}; shift || true
args="--steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-extras --streams 1 --profile-steps 0 $*"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $args > $out/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $args > $out/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $args > $out/write.log 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
  -d $out/sq -- python3 bench.py $args > $out/sq.log 2>&1
echo "sq pass done"
python3 profiles/summarize.py stats $out/stats $out/stats_table.md > /dev/null
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py $args"; echo; cat $out/stats_table.md; } > profiles/${tag}_kernel_stats.md
python3 profiles/summarize.py pmc $out/fetch $out/write profiles/${tag}_pmc.json "bench.py $args" > /dev/null
python3 profiles/summarize.py sq $out/sq profiles/${tag}_sq_counters.json > /dev/null
grep '^{' $out/stats.log | tail -1 > profiles/${tag}_bench_under_rocprof.json || true
# per-class HBM bytes for bench.py's roofline.traffic (tagged there as coming from these passes, not from the bench run itself)
prec=f16x3; case "$*" in *fp32*) prec=fp32;; *bf16*) prec=bf16;; esac
T=243; B=64; prev=""
for w in "$@"; do
  [ "$prev" = "--frames" ] && T=$w
  [ "$prev" = "--batch" ] && B=$w
  prev=$w
done
python3 profiles/summarize.py traffic profiles/${tag}_pmc.json profiles/hbm_traffic.json $T $B $prec > /dev/null || true
python3 - <<PY
import json, time
import hashlib
p = "profiles/hbm_traffic.json"; d = json.load(open(p)); d["_collected"] = "${tag}, " + time.strftime("%Y-%m-%d")
# the library these bytes were counted with, per workload: bench.py prints roofline.traffic_stale when the running library differs
d.setdefault("_lib_sha256", {})["T${T}:B${B}:${prec}"] = hashlib.sha256(open("diff3dhpe_amd/libd3d_hip.so", "rb").read()).hexdigest()
json.dump(d, open(p, "w"), indent=1)
PY
# one reviewable table per tag: launch time x algorithmic work x counters (add the sustained MFMA rate of bench.py's machine_probes by hand:
#   python3 profiles/roofline_table.py $tag $T $B $prec <TFLOP/s>)
python3 profiles/roofline_table.py $tag $T $B $prec > profiles/${tag}_roofline.md || true
cp profiles/${tag}_* profiles/hbm_traffic.json gpurun_out/ 2>/dev/null || true
cat profiles/${tag}_kernel_stats.md
