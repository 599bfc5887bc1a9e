#!/bin/bash
# timeline of the headline's steps (rocprofv3 --kernel-trace of a short bench run): which kernel ran when, on which queue
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-tr4}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
cd /tmp
PSGPU_BENCH_NO_PCIE=1 PSGPU_BENCH_PIPES=${2:-2} timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tr -o tr -- python $R/bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench.log 2>&1
cd $R
python tools/overlap_trace.py $OUT/tr > $OUT/timeline.txt 2>&1
grep -n "timeline" -A200 $OUT/timeline.txt | head -150
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
