#!/bin/bash
# A/B of library variants on the driver's bench command: tools/ab_generic.sh "<kernel names>" <variant.so> [...]; the in-tree build is "base"
K="$1"; shift
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-34s' % sys.argv[1], d['value'], d['ms_per_step'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in sys.argv[2].split() if n in k))" "$1" "$K"; }
run() { timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  run base
  for L in "$@"; do GSR_LIB=$PWD/$L run $L; done
done
