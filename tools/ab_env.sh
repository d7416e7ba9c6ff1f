#!/bin/bash
# A/B of environment switches on the driver's bench command:  tools/ab_env.sh VAR=a VAR=b ...   (GPU box, repo root;
# AB_CONFIG=N picks another BASELINE config).  Two interleaved rounds to see the run-to-run spread.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
print('%-22s' % sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median'], 'fwd', d['fwd_ms'], 'ksum', d['kernel_ms_sum'], ' '.join('%s %.4f' % (n[:14], k[n]['avg_ms']) for n in k if any(s in n for s in ('emit','radix','digit','scan_block','preprocess_fwd','finalize','tile_'))))" "$1"; }
CFG=${AB_CONFIG:-3}
for i in 1 2; do
  for kv in "$@"; do
    env $kv BENCH_C5_STATIC=1 timeout -k 10 200 python bench.py --config $CFG --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null && show $kv
  done
done
