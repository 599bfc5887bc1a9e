#!/bin/bash
# A/B of library variants on the large-vocabulary leg (one box): B = 256, 2 steps each, alternating
set -u
TAG=${1:-abl}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  PSGPU_LIB_PATH=$L timeout 600 python bench.py --workload large --steps 2 --no-cpu-baseline --utts 256 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', j['value'], j['ms_per_step'], j['stage_ms']['search'])" | tee -a "$OUT/ab.txt"
done
