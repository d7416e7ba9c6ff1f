#!/bin/bash
# PMC passes over the one-wave-per-tile compositing backward: sums on the matrix pipe (GSR_BWD_REDUCE=mfma, k_render_bwd_tile_mx)
# against the v_permlane / DPP tree (swap):   tools/pmc_bwd_reduce.sh <outdir> [lib.so]    (repo root, GPU box; counters only)
set -e
OUT=$1; shift
R=$PWD
[ -n "$1" ] && export GSR_LIB=$R/$1
export TMPDIR=/tmp
mkdir -p $R/$OUT
cd /tmp
for red in mfma swap; do
  n=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    n=$((n+1))
    GSR_BWD_REDUCE=$red rocprofv3 --kernel-trace --pmc $set \
      --kernel-include-regex "k_render_bwd" --output-format csv -d $R/$OUT/pmc_$red$n -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 8 > $R/$OUT/pmc_$red$n.json 2> $R/$OUT/pmc_$red$n.err || echo "pass $red $n failed"
  done
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections, statistics as st
out = sys.argv[1]
for red in ("mfma", "swap"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{red}*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"GSR_BWD_REDUCE={red}:", {k: f"{st.mean(v):.4g}" for k, v in sorted(agg.items())}, "dispatches", max(len(v) for v in agg.values()) if agg else 0)
PY
