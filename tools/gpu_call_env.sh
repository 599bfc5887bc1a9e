#!/bin/bash
# headline line under environment settings: tools/gpu_call_env.sh TAG "VAR=val VAR2=val" ...
set -u
TAG=${1:-env}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for e in "$@"; do
  echo "== $e" | tee -a "$OUT/bench.txt"
  env PSGPU_BENCH_NO_PCIE=1 $e timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee -a "$OUT/bench.txt"
done
tail -3 "$OUT/bench.err"
