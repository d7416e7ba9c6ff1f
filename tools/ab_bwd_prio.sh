#!/bin/bash
# A/B of issue priorities for long walks in the one-wave-per-tile compositing backward (GSR_BWD_PRIO="t1,t2,t3"), driver command.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-34s' % sys.argv[1], d['value'], d['ms_per_step'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','preprocess_bwd_adam') if n in k))" "$1"; }
run() { timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "${@:2}" > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  run "no priorities"
  for t in "230,270,310" "210,250,290" "250,300,340" "205,206,207" "150,250,350" "260,261,262"; do
    GSR_BWD_PRIO=$t run "prio $t"
  done
done
