#!/bin/bash
# tools/experiments/run_ab.sh <outdir-under-gpurun_out>: the per-Gaussian backward A/B, its rocprofv3 kernel stats and one PMC pass
OUT=$1; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/experiments/bwd_gauss_ab.py > $OUT/ab_plain.json 2> $OUT/ab_plain.err; cat $OUT/ab_plain.json
BG_LIB=libbwd_gauss_compact.so python3 tools/experiments/bwd_gauss_ab.py > $OUT/ab_compact.json 2> $OUT/ab_compact.err; cat $OUT/ab_compact.json
cd /tmp
BG_LIB=libbwd_gauss_compact.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/tools/experiments/bwd_gauss_ab.py > /dev/null 2> $R/$OUT/stats.err
BG_LIB=libbwd_gauss_compact.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$OUT/pmc1 -- python3 $R/tools/experiments/bwd_gauss_ab.py --reps 3 > /dev/null 2> $R/$OUT/pmc1.err
echo done
