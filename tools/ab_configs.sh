#!/bin/bash
# A/B of library builds over the BASELINE configs (driver command form): tools/ab_configs.sh <lib.so> [configs...]
L=$1; shift; CFGS=${@:-1 2 3 4 5}
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-10s c%s' % (sys.argv[1], sys.argv[2]), d['value'], d['ms_per_step'], 'fwd', d['fwd_ms'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n[:12], k[n]['avg_ms']) for n in k if any(s in n for s in ('tile_','radix','emit','finalize'))))" "$1" "$2"; }
for c in $CFGS; do
  for i in 1 2; do
    BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config $c --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show base $c
    BENCH_C5_STATIC=1 GSR_LIB=$PWD/$L timeout -k 10 200 python bench.py --config $c --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show variant $c
  done
done
