#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs) + kernel stats of the large-vocabulary leg at N utterances
set -u
TAG=${1:-lvpmc}; N=${2:-256}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH="python $PWD/bench.py --workload large --steps 1 --no-cpu-baseline --utts $N --large-vocab-utts $N"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1
cd - > /dev/null
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
python tools/prof_collect.py "$OUT" "$TAG" $N 30.0 decode_large_vocab 2>&1 | grep -A9 '"fwdtree_kernel"'
