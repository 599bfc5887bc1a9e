#!/bin/bash
# round 4, call a: the GPU suite on the round's first commit, then the search alone at 256 / 128 / 64 work-items per
# utterance (A/B libraries built with -DPSGPU_FT_THREADS) and the phase profile
set -u
TAG=${1:-r4a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
for v in default nt128 nt64; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/ab.txt"
  PSGPU_LIB_PATH=$L SB_BATCHES=512,768,1024 SB_REPS=5 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/ab.txt"
done
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so SB_BATCHES=512 SB_REPS=1 timeout 300 python tools/search_bench.py > "$OUT/prof.txt" 2>&1
tail -45 "$OUT/prof.txt"
