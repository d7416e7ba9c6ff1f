#!/bin/bash
# A/B of the two binning forms on the driver's bench command (run on the GPU box from the repo root):
#   tile-local depth ordering (GSR_BINNING=tile) against the global depth sort (GSR_BINNING=global).
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/tlo.json').read()); k=d['kernels']
print(sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median'], 'fwd', d['fwd_ms'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('tile_depth_sort','tile_depth_sort_long','emit_instances') if n in k))" $1; }
for i in 1 2; do
  GSR_BINNING=tile timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tlo.json 2>/dev/null && show tile
done
GSR_BINNING=global timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tlo.json 2>/dev/null && show global
