#!/bin/bash
# A/B: first two stages of the backward's wave reduction through LDS (in-tree) against the v_permlane swap tree (variant nolds =
# -DBWD_LDS_REDUCE=0) and the round-3 loop (GSR_BWD_MASK=0); driver command, interleaved twice.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-34s' % sys.argv[1], d['value'], d['ms_per_step'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','preprocess_bwd_adam') if n in k))" "$1"; }
run() { timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "${@:2}" > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  run "LDS stages (in-tree)"
  GSR_LIB=$PWD/tools/variants/libgsr_nolds.so run "swap tree (nolds)"
  GSR_BWD_MASK=0 run "round-3 loop (GSR_BWD_MASK=0)"
done
