#!/bin/bash
# A/B of libraries on the search alone (turtle, LDS layout): tools/gpu_call_ab2.sh TAG variant...
set -u
TAG=${1:-ab}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/ab.txt"
  PSGPU_LIB_PATH=$L SB_BATCHES=512 SB_REPS=8 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/ab.txt"
done
