#!/bin/bash
# A/B several libgsr_hip.so builds in ONE process-sequence on one GPU box: tools/ab_bench.sh base.so varA.so varB.so ...
# prints it/s and the per-kernel times named in $AB_KERNELS (default: the render kernels), interleaved twice to see the
# run-to-run spread
KERNELS=${AB_KERNELS:-render_bwd render_fwd}
for round in 1 2; do
  for lib in "$@"; do
    GSR_LIB=$lib python bench.py --steps 40 --warmup 8 --views 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('%-24s %7.1f it/s  %.3f ms/step  ' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step']) + '  '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in '$KERNELS'.split() if n in k))"
  done
done
