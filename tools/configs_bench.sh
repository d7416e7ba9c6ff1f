#!/bin/bash
# BASELINE configs 1, 2, 4, 5 (static) through the driver's command form, one JSON line each into gpurun_out/$1/
out=gpurun_out/${1:-cfg}; mkdir -p $out
for c in 1 2 4; do
  timeout -k 10 200 python bench.py --config $c --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/c$c.json 2> $out/c$c.err || echo "config $c failed"
  python tools/show_bench.py $out/c$c.json tile radix emit render
done
BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config 5 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/c5.json 2> $out/c5.err || echo "config 5 failed"
python tools/show_bench.py $out/c5.json tile radix emit render
