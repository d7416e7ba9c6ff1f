#!/bin/bash
# A/B of the validity flags of the per-instance gradient records against the build before them (tools/variants/libgsr_before_flags.so);
# driver command; C3, 2 x splats, C4, C2
row() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-8s %-13s' % (sys.argv[2], sys.argv[1]), d['value'], d['ms_per_step'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','preprocess_bwd_adam') if n in k))" "$1" "$2"; }
for cfg in "c3:" "heavy:--scale-factor 2" "c4:--config 4 --views 4" "c2:--config 2"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  for i in 1 2; do
    for L in before_flags base; do
      if [ $L = base ]; then unset GSR_LIB; else export GSR_LIB=$PWD/tools/variants/libgsr_$L.so; fi
      timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $extra > gpurun_out/ab.json 2>/dev/null && row $L $name
    done
  done
done
unset GSR_LIB
