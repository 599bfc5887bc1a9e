#!/bin/bash
# A/B of library variants on one box: search alone (golden replicas) and the headline bench
set -u
TAG=${1:-ab}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/ab.txt"
  PSGPU_LIB_PATH=$L SB_BATCHES=512,768 SB_REPS=5 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/ab.txt"
  PSGPU_LIB_PATH=$L timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])" | tee -a "$OUT/ab.txt"
done
