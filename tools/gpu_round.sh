#!/bin/bash
# tools/gpu_round.sh -- one gpurun call: GPU parity tests, bench line, rocprofv3 kernel-trace stats and the two PMC
# passes of the HEADLINE workload alone (FETCH_SIZE / WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC
# slots"; PMC never together with a trace domain other than --kernel-trace).  Everything lands under gpurun_out/<tag>/;
# tools/prof_collect.py condenses it for profiles/.
#   usage: tools/gpu_round.sh TAG [skip-tests|-] [search] [noprof]
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0

if [ "${2:-}" != "skip-tests" ]; then
  (timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40) > "$OUT/pytest_gpu.log"
  cat "$OUT/pytest_gpu.log"
  (timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3) > "$OUT/smoke.log"
  cat "$OUT/smoke.log"
fi

timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 6000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"

if [ "${4:-}" != "noprof" ]; then
  BENCH="python $PWD/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1
  cd - > /dev/null
  find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
  python tools/prof_collect.py "$OUT" "$TAG" 512 30.0 2>&1 | tail -60
fi
# search kernel alone (golden replicas: the reference's scores as inputs), both layouts on the small task
if [ "${3:-}" = "search" ]; then
  SB_BATCHES=1,512,1024,2048 timeout 300 python tools/search_bench.py > "$OUT/search_turtle_lds.txt" 2>&1
  PSGPU_FWDTREE_LAYOUT=slab SB_BATCHES=512,1024 timeout 300 python tools/search_bench.py > "$OUT/search_turtle_slab.txt" 2>&1
  SB_CASE=medium_goforward SB_BATCHES=1,512,1024 timeout 300 python tools/search_bench.py > "$OUT/search_medium.txt" 2>&1
  SB_CASE=cmudict SB_BATCHES=1,32,256 SB_REPS=1 timeout 600 python tools/search_bench.py > "$OUT/search_cmudict.txt" 2>&1
  tail -5 "$OUT"/search_*.txt
fi
