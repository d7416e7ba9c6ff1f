#!/bin/bash
# A/B of a variant library against the in-tree build on the other BASELINE configs (GPU box, repo root):
#   tools/ab_cfg_lib.sh tools/variants/libgsr_<name>.so [configs, default "1 2 5 4"]      interleaved, variant first
for c in ${2:-1 2 5 4}; do
  for i in 1 2; do
    for L in "$1" ""; do
    GSR_LIB=${L:+$PWD/$L} BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config $c --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/ab.json 2>/dev/null && python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print('C$c', sys.argv[1] or 'in-tree', d['value'], d['ms_per_step'], d['ms_per_step_median'])" "$L"
    done
  done
done
