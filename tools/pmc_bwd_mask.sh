#!/bin/bash
# PMC pass over the one-wave-per-tile compositing backward, with and without the round-4 sub-block masks:
#   tools/pmc_bwd_mask.sh <outdir>     (run from the repo root on the GPU box)
set -e
OUT=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/$OUT
cd /tmp
for mask in 1 0; do
  GSR_BWD_MASK=$mask rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
    --kernel-include-regex "k_render_bwd" --output-format csv -d $R/$OUT/pmc_mask$mask -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 8 > $R/$OUT/pmc_mask$mask.json 2> $R/$OUT/pmc_mask$mask.err
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections, statistics as st
out = sys.argv[1]
for mask in (1, 0):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_mask{mask}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"GSR_BWD_MASK={mask}:", {k: f"{st.mean(v):.4g}" for k, v in sorted(agg.items())}, "dispatches", max(len(v) for v in agg.values()) if agg else 0)
PY
