#!/bin/bash
# round 4, call b: the rest of the GPU suite; the large-vocabulary search at 1024 / 768 / 512 work-items per utterance
# (A/B libraries, -DPSGPU_FT_THREADS_BIG); phase profile of the LDS layout at 64 work-items
set -u
TAG=${1:-r4b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_zz_flat_gpu.py tests/test_zz_search_layouts_gpu.py -m gpu -q -x 2>&1 | tail -8) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
for v in default big768 big512; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/ab_big.txt"
  PSGPU_LIB_PATH=$L SB_CASE=cmudict SB_BATCHES=32,256 SB_REPS=2 timeout 600 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/ab_big.txt"
done
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof64.so SB_BATCHES=512 SB_REPS=1 timeout 300 python tools/search_bench.py > "$OUT/prof64.txt" 2>&1
tail -42 "$OUT/prof64.txt"
