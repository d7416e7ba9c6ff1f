#!/bin/bash
# A/B of the round-4 binning chain (histograms counted by the emission kernel, tile ranges from the sort's last pass: no
# k_radix_hist_all, no k_finalize_bins) against the round-3 chain (GSR_TILE_HIST=0); driver command, interleaved twice.
show() { python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read()); k=d['kernels']
names=('preprocess_fwd','emit_instances','radix_hist','radix_pass','finalize_bins','tile_depth_sort','tile_depth_sort_long')
tot=sum(k[n]['avg_ms']*k[n]['calls']/max(1,k['emit_instances']['calls']) for n in names if n in k)
print('%-30s' % sys.argv[1], d['value'], d['ms_per_step'], 'chain %.4f' % tot, ' '.join('%s %.4f' % (n, k[n]['avg_ms']) for n in names if n in k))" "$1"; }
run() { timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "${@:2}" > gpurun_out/ab.json 2>/dev/null && show "$1"; }
for i in 1 2; do
  run "fused chain (in-tree)"
  GSR_TILE_HIST=0 run "round-3 chain (GSR_TILE_HIST=0)"
done
