#!/bin/bash
# A/B of the compositing backward's sub-block masks (round 4) on the driver's bench command, one GPU box:
#   in-tree build (masks, 6 waves/SIMD forced: 80 VGPRs + spills in the staging code), the same without forcing (variants w5 / w4
#   from tools/mkvariant.sh), and the round-3 loop (GSR_BWD_MASK=0).  Interleaved twice.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-34s' % sys.argv[1], d['value'], d['ms_per_step'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','preprocess_bwd_adam') if n in k))" "$1"; }
run() { timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "${@:2}" > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  run "in-tree"
  GSR_FWD_MASK=0 run "forward without masks (GSR_FWD_MASK=0)"
  GSR_BWD_MASK=0 run "round-3 loop (GSR_BWD_MASK=0)"
  for v in none; do
    [ -f tools/variants/libgsr_$v.so ] && GSR_LIB=$PWD/tools/variants/libgsr_$v.so run "mask $v"
  done
done
