#!/bin/bash
# Rehearsal of the N > 1 training schedule on a ONE-GPU box over real RCCL: a process group of one rank, every exchange form,
# with and without the side-stream overlap (GPU box, repo root).  What it shows: every collective call of scene_utils/parallel.py is
# accepted and executed by RCCL (dtypes, shapes, in-place forms, AVG / MAX ops, async work on a side stream), the schedule does not
# deadlock, and what the extra kernels of the N > 1 schedule cost per step.  What it cannot show: link time.
mkdir -p gpurun_out
for ex in sh_rank1 allreduce visible_rows sharded; do
  for ov in "" "--no-overlap"; do
    [ "$ex" = sharded ] && [ -z "$ov" ] && continue
    opt=hip_fused; [ "$ex" = visible_rows ] && opt=hip_sparse; [ "$ex" = sharded ] && opt=hip
    BENCH_SINGLE_RANK_GROUP=1 timeout -k 10 240 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile \
      --exchange $ex --optimizer $opt $ov > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err || { echo "FAILED $ex $ov"; tail -20 gpurun_out/rehearse.err; exit 1; }
    python3 -c "
import json,sys;d=json.loads(open('gpurun_out/rehearse.json').read().strip().splitlines()[-1]);c=d['config']
print('%-13s %-12s' % (sys.argv[1], sys.argv[2] or 'overlap'), d['value'], 'it/s', d['ms_per_step'], 'ms  overlap', c['overlap_comm'], 'exchange', c['exchange'], 'optimizer', c.get('optimizer'))" "$ex" "$ov"
  done
done
