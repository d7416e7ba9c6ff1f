#!/bin/bash
# A/B of library builds on the driver's bench command:  tools/ab_lib.sh <lib.so> [<lib.so> ...]   (GPU box, repo root; the
# in-tree build is measured as "base" beside every variant)
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-28s' % sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','tile_depth_sort','radix_pass','radix_hist','emit_instances') if n in k))" "$1"; }
for i in 1 2; do
  timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show base
  for L in "$@"; do
    GSR_LIB=$PWD/$L timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show $L
  done
done
