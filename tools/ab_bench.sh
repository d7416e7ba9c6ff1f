#!/bin/bash
# A/B several libgsr_hip.so builds in ONE process-sequence on one GPU box: tools/ab_bench.sh base.so varA.so varB.so ...
# prints render kernel times and it/s for each (interleaved twice to see run-to-run spread)
for round in 1 2; do
  for lib in "$@"; do
    GSR_LIB=$lib python bench.py --steps 40 --warmup 8 --views 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('%-34s %7.1f it/s  %.3f ms/step  bwd %.4f fwd %.4f' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step'], k['render_bwd']['avg_ms'], k['render_fwd']['avg_ms']))"
  done
done
