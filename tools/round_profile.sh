#!/bin/bash
# Collects the evidence of a round on the GPU box:  tools/round_profile.sh <outdir-under-gpurun_out> [tag]
#   bench lines of every BASELINE config, rocprofv3 --kernel-trace --stats of the driver's command, PMC passes (separate runs:
#   SQ counters, FETCH_SIZE, WRITE_SIZE - /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Run from the repo root; summaries are copied into profiles/ afterwards (tools/pmc_summary.py for the PMC passes).
set -x
OUT=$1
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5"
timeout -k 10 300 python3 bench.py $DRV > $OUT/bench_c3_driver.json 2> $OUT/bench_c3_driver.err
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_c3_50.json 2> $OUT/bench_c3_50.err
timeout -k 10 200 python3 bench.py $DRV --config 1 --cpu-tiles 256 > $OUT/bench_c1.json 2> $OUT/bench_c1.err
timeout -k 10 200 python3 bench.py $DRV --config 2 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -k 10 300 python3 bench.py $DRV --config 4 --views 4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
# BASELINE configs[4]: the preset grows 50 k -> 500 k (bench.py: --config 5 = reference schedule from iteration 100, threshold 2e-5, stop at 500 k, 2500 steps)
timeout -k 10 400 python3 bench.py --config 5 --warmup 10 --no-cpu-baseline > $OUT/bench_c5_growth500k.json 2> $OUT/bench_c5_growth500k.err
timeout -k 10 300 python3 bench.py --config 5 --densify --densify-from 100 --densify-grad-threshold 0.0002 --steps 2500 --warmup 10 --no-cpu-baseline > $OUT/bench_c5_reference_threshold.json 2> $OUT/bench_c5_reference_threshold.err
# NOT a BASELINE config: C3 with twice the recipe's splat size - the compositing kernels dominate as on a real capture
timeout -k 10 300 python3 bench.py $DRV --scale-factor 2 --no-cpu-baseline > $OUT/bench_c3_heavy.json 2> $OUT/bench_c3_heavy.err
timeout -k 10 200 python3 bench.py $DRV --no-cpu-baseline --optimizer hip > $OUT/bench_c3_adam_unfused.json 2> /dev/null
timeout -k 10 200 python3 bench.py $DRV --no-cpu-baseline --optimizer hip_sparse_fused > $OUT/bench_c3_sparse_fused.json 2> /dev/null
timeout -k 10 200 python3 bench.py $DRV --no-cpu-baseline --forward-mode sync > $OUT/bench_c3_sync_forward.json 2> /dev/null
timeout -k 10 200 python3 bench.py $DRV --no-cpu-baseline --forward-mode async > $OUT/bench_c3_async_forward.json 2> /dev/null
GSR_BINNING=global timeout -k 10 200 python3 bench.py $DRV --no-cpu-baseline > $OUT/bench_c3_global_binning.json 2> /dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py $DRV --no-cpu-baseline > $R/$OUT/stats.json 2> $R/$OUT/stats.err
PM="--gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 8"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$OUT/pmc1 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc1.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc2 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc2.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc3 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc3.err
echo done
