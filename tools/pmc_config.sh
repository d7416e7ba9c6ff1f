#!/bin/bash
# PMC passes (SQ counters / FETCH_SIZE / WRITE_SIZE, separate runs) of one BASELINE config:
#   tools/pmc_config.sh <outdir-under-gpurun_out> <config> [extra bench args]      (from the repo root, on the GPU box)
# then, back in the container:  python tools/pmc_summary.py r02_pmc_c<config> c<config>:<P>:<W>x<H> <outdir>/pmc1 <outdir>/pmc2 <outdir>/pmc3
set -x
OUT=$1; CFG=$2; shift; shift
R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PM="--gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 4 --config $CFG $@"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$OUT/pmc1 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc1.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc2 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc2.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc3 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc3.err
echo done
