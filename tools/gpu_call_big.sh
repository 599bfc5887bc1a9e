#!/bin/bash
# the large-vocabulary search alone (134,865 words, trace made on the spot by the compiled reference): tools/gpu_call_big.sh TAG variant...
set -u
TAG=${1:-big}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/ab_big.txt"
  PSGPU_LIB_PATH=$L SB_CASE=cmudict SB_BATCHES=${BIGB:-256} SB_REPS=2 timeout 600 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/ab_big.txt"
done
