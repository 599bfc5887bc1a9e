#!/bin/bash
# round 6, call r: the second pass's time against the number of utterances in flight (what shares a compute unit's caches)
set -u
TAG=${1:-r6_r}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 32 64 128 256 512; do
  echo "== B=$b" | tee -a "$OUT/out.txt"
  TPP_B=$b timeout 600 python tools/two_pass_pipeline_prof.py 2>> "$OUT/plain.err" | tee -a "$OUT/out.txt"
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TPP_B=$b TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py 2>&1 >/dev/null | grep "fwdflat_kernel profile\|slowest" | tail -2 | tee -a "$OUT/out.txt"
done
