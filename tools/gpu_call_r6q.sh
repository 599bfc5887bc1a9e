#!/bin/bash
# round 6, call q: second-pass variants side by side -- plain step times (twice each) and the phase profile of listed builds
# usage: gpu_call_r6q.sh TAG "plainlib1 plainlib2 ..." "proflib1 ..."   (library names without the libpsgpu_ prefix; "default" = libpsgpu.so)
set -u
TAG=${1:-r6_q}; PL=${2:-default}; PR=${3:-prof}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
lib() { [ "$1" = "default" ] && echo $PWD/pocketsphinx_amd/libpsgpu.so || echo $PWD/pocketsphinx_amd/libpsgpu_$1.so; }
for v in $PL $PL; do
  echo "== $v" | tee -a "$OUT/plain.txt"
  PSGPU_LIB_PATH=$(lib $v) timeout 600 python tools/two_pass_pipeline_prof.py 2>> "$OUT/plain.err" | tee -a "$OUT/plain.txt"
done
for v in $PR; do
  echo "== $v" | tee -a "$OUT/prof.txt"
  PSGPU_LIB_PATH=$(lib $v) TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py > /dev/null 2> "$OUT/prof_$v.err"
  grep -A24 "fwdflat_kernel profile" "$OUT/prof_$v.err" | tail -25 | tee -a "$OUT/prof.txt"
done
