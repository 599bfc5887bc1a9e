#!/bin/bash
# round 6, call u: the C batch call after a change -- its GPU tests, then the three-pass leg (128 x 30 s)
set -u
TAG=${1:-r6_u}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 2400 python -m pytest tests/test_dropin_gpu.py tests/test_zz_asan_gpu.py -m gpu -x -q -k "batch or asan" 2>&1 | tail -6) | tee "$OUT/pytest.log"
for b in 128 512; do TP3_B=$b timeout 900 python tools/three_pass_bench.py 2> "$OUT/tp3.err" | tee -a "$OUT/tp3.json"; done
