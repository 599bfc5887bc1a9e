#!/bin/bash
# phase profile of the tree search (LDS layout, golden replicas): tools/gpu_call_prof.sh TAG [lib]
set -u
TAG=${1:-prof}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/${2:-libpsgpu_prof.so} SB_BATCHES=512 SB_REPS=1 timeout 300 python tools/search_bench.py > "$OUT/prof.txt" 2>&1
tail -42 "$OUT/prof.txt"
