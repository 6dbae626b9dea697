#!/bin/bash
# experiments/lds_conflict_attribution.sh NAME...   (on the GPU box): one rocprofv3 --pmc pass (LDS cycles / bank-conflict cycles) of a
# short one-stream bench per library variant experiments/_libs/libd3d_NAME.so ("cur" = the in-tree build); the -DQS_ABL / -DQT_ABL builds
# of the two fused kernels (wrong results, same control flow) tell which LDS accesses the conflict cycles belong to.
set -eo pipefail
cur=diff3dhpe_amd/libd3d_hip.so
cp $cur /tmp/_lib_cur.so
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = cur ]; then cp /tmp/_lib_cur.so $cur; else cp experiments/_libs/libd3d_$n.so $cur; fi
  out=gpurun_out/ldsattr_$n; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 0 \
    --no-cpu-baseline --no-selfcheck --no-extras --streams 1 --profile-steps 0 > $out/run.log 2>&1 || { cp /tmp/_lib_cur.so $cur; tail -5 $out/run.log; exit 1; }
  python3 profiles/summarize.py sq $out $out/sq.json > /dev/null
  python3 - "$n" $out/sq.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
for k, v in d.items():
    if "k_qkv" in k:
        print("%-10s %-12s LDS active %.4g  conflict %.4g  ratio %.3f  launches %s" % (sys.argv[1], k[:11], v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v.get("launches")))
PY
done
cp /tmp/_lib_cur.so $cur
