#!/bin/bash
# A/B of the compositing backward's reduction: matrix pipe (opt-in, k_render_bwd_tile_mx) against the default v_permlane / DPP tree
# (GSR_BWD_REDUCE=swap); driver command, interleaved twice.  AB_CONFIG=N / AB_EXTRA="--scale-factor 2" pick another workload.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-28s' % sys.argv[1], d['value'], d['ms_per_step'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','preprocess_bwd_adam','preprocess_bwd') if n in k))" "$1"; }
CFG=${AB_CONFIG:-3}
for i in 1 2; do
  for kv in GSR_BWD_REDUCE=mfma GSR_BWD_REDUCE=swap; do
    env $kv BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config $CFG --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $AB_EXTRA > gpurun_out/ab.json 2>/dev/null && show $kv
  done
done
