#!/bin/bash
# round 6, call p: the second pass after a change -- its GPU tests, the plain step times, the phase profile
set -u
TAG=${1:-r6_p}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests/test_zz_flat_gpu.py tests/test_decode_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -5) | tee "$OUT/pytest.log"
for i in 1 2; do timeout 600 python tools/two_pass_pipeline_prof.py 2> "$OUT/plain.err" | tee -a "$OUT/plain.json"; done
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py > "$OUT/prof.json" 2> "$OUT/prof.err"
grep -A20 "fwdflat_kernel profile" "$OUT/prof.err" | tail -21
grep "fwdflat host" "$OUT/prof.err" | tail -1
