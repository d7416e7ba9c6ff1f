#!/bin/bash
# A/B of one environment switch on the driver's bench command:  tools/ab_env.sh VAR=VALUE [VAR=VALUE ...]   (GPU box, repo root)
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print(sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median'], 'fwd', d['fwd_ms'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('tile_depth_sort','emit_instances','shade','render_fwd') if n in k))" "$1"; }
for i in 1 2; do
  timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show base
  env "$@" timeout -k 10 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show "$*"
done
