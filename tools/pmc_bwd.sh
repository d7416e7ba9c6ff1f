#!/bin/bash
# PMC pass over the compositing kernels: tools/pmc_bwd.sh <outdir> [env...]   (run from the repo root on the GPU box)
set -e
OUT=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
for form in tile quad; do
  GSR_BWD_FORM=$form rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
    --kernel-include-regex "k_render" --output-format csv -d $R/$OUT/pmc_$form -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --views 8 > $R/$OUT/pmc_$form.json 2> $R/$OUT/pmc_$form.err
done
