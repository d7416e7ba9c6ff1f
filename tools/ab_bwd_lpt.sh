#!/bin/bash
# A/B of the walk classes (k_render_bwd_tile takes the tiles longest walk first): default against GSR_BWD_LPT=0 (index order) and a
# variant build (tools/variants/libgsr_sub2.so, e.g. tools/mkvariant.sh sub2 render.hip "-DGSR_WALK_SUB=1"); driver command.  AB_CONFIG / AB_EXTRA as usual.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-26s' % sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median'], ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in ('render_bwd','render_fwd','tile_depth_sort','preprocess_bwd_adam') if n in k))" "$1"; }
run() { BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config ${AB_CONFIG:-3} --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $AB_EXTRA > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  GSR_BWD_LPT=0 run "index order"
  run "walk classes (default build)"
  [ -f tools/variants/libgsr_sub2.so ] && GSR_LIB=$PWD/tools/variants/libgsr_sub2.so run "walk classes (variant build)"
done
