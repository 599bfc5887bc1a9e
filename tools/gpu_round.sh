#!/bin/bash
# tools/gpu_round.sh -- one gpurun call: GPU parity tests, bench line, rocprofv3
# kernel-trace stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE cannot
# share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Everything lands
# under gpurun_out/<tag>/; tools/prof_collect.py condenses it for profiles/.
#   usage: tools/gpu_round.sh TAG [skip-tests|-] [search]
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0

if [ "${2:-}" != "skip-tests" ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > "$OUT/pytest_gpu.log"
  cat "$OUT/pytest_gpu.log"
  (timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3) > "$OUT/smoke.log"
  cat "$OUT/smoke.log"
fi

timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"; tail -3 "$OUT/bench.err"

BENCH="python $PWD/bench.py --no-cpu-baseline --steps 20 --warmup 3"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1
cd - > /dev/null
# search kernels: both formulations of the tree search on the small and the full-cmudict task, the flat pass
if [ "${3:-}" = "search" ]; then
  for mode in per_node active_list; do
    SB_MODE=$mode SB_BATCHES=1,512,1024 timeout 300 python tools/search_bench.py > "$OUT/search_turtle_$mode.txt" 2>&1
    SB_MODE=$mode SB_CASE=medium_goforward SB_BATCHES=1,512 timeout 300 python tools/search_bench.py > "$OUT/search_medium_$mode.txt" 2>&1
    SB_MODE=$mode SB_CASE=cmudict SB_BATCHES=1,32,256 SB_REPS=1 timeout 600 python tools/search_bench.py > "$OUT/search_cmudict_$mode.txt" 2>&1
  done
  FB_BATCHES=1,64,512 timeout 300 python tools/flat_bench.py > "$OUT/flat_bench.txt" 2>&1
  tail -4 "$OUT"/search_*.txt "$OUT/flat_bench.txt"
fi
# keep only what is small enough to merge back
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
python tools/prof_collect.py "$OUT" "$TAG" 2>&1 | tail -40
