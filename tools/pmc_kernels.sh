#!/bin/bash
# One PMC pass (8 SQ counters, kernel trace only) over the kernels matching a regex, for the in-tree build and any variants:
#   tools/pmc_kernels.sh <outdir> "<kernel regex>" "<counter list>" [variant.so ...]     (run from the repo root on the GPU box)
set -e
OUT=$1; RE=$2; CTR=$3; shift 3
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/$OUT
cd /tmp
n=0
for L in base "$@"; do
  n=$((n+1))
  if [ "$L" = base ]; then unset GSR_LIB; else export GSR_LIB=$R/$L; fi
  rocprofv3 --kernel-trace --pmc $CTR --kernel-include-regex "$RE" --output-format csv -d $R/$OUT/pmc_$n -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 8 > $R/$OUT/pmc_$n.json 2> $R/$OUT/pmc_$n.err
done
cd $R
python3 - $OUT base "$@" <<'PY'
import csv, glob, sys, collections, statistics as st
out, libs = sys.argv[1], sys.argv[2:]
for n, lib in enumerate(libs, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_{n}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        print(lib, k, {a: f"{st.mean(v):.4g}" for a, v in sorted(c.items())})
PY
