#!/bin/bash
# occupancy experiment of the LDS-layout search kernel: variants x batch sizes (search alone), then the bench with the best
set -u
TAG=${1:-occ}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "" r0w4 r1w3 r1w4; do
  L=$PWD/pocketsphinx_amd/libpsgpu${v:+_$v}.so
  echo "== variant ${v:-default}" | tee -a "$OUT/search.txt"
  PSGPU_LIB_PATH=$L SB_BATCHES=512,768,1024,1536 SB_REPS=3 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/search.txt"
done
for v in "" r1w4; do
  L=$PWD/pocketsphinx_amd/libpsgpu${v:+_$v}.so
  echo "== bench variant ${v:-default}" | tee -a "$OUT/bench.txt"
  PSGPU_LIB_PATH=$L timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])" | tee -a "$OUT/bench.txt"
done
