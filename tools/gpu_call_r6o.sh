#!/bin/bash
# round 6, call o: the second pass's phase profile with the word transitions' shape (profile build), and the plain step times
set -u
TAG=${1:-r6_o}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/two_pass_pipeline_prof.py > "$OUT/plain.json" 2> "$OUT/plain.err"
cat "$OUT/plain.json"
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py > "$OUT/prof.json" 2> "$OUT/prof.err"
grep -A20 "fwdflat_kernel profile" "$OUT/prof.err" | tail -21
grep "fwdflat host" "$OUT/prof.err" | tail -1
