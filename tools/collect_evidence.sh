#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/collect_evidence.sh <tag>): default bench line, rocprofv3 kernel stats of the same
# command, and four separate PMC passes (SQ x2, FETCH_SIZE, WRITE_SIZE - never combined with other traces).  Everything lands
# under gpurun_out/<tag>_*; tools/pmc_summary.py turns the PMC passes into profiles/*.csv afterwards.
set -o pipefail
TAG=${1:-ev}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_default.json 2> $R/gpurun_out/${TAG}_bench_default.err || exit 1
echo bench=$(cut -c1-120 $R/gpurun_out/${TAG}_bench_default.json)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 30 --warmup 5 --views 8 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.json 2> $R/gpurun_out/${TAG}_stats.err || exit 2
echo stats=ok
B="python3 $R/bench.py --steps 6 --warmup 2 --views 4 --no-cpu-baseline --no-kernel-profile"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_pmc1 -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc1.err || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/${TAG}_pmc2 -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc2.err || exit 4
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc3 -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc3.err || exit 5
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc4 -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc4.err || exit 6
echo pmc=ok
