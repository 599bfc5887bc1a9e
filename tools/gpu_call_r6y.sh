#!/bin/bash
# round 6, call y: the second pass's launch order (largest utterances first, paired with the smallest) on / off: PSGPU_FF_ORDER
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0 1 0; do
  echo "== PSGPU_FF_ORDER=$v"
  PSGPU_FF_ORDER=$v timeout 600 python tools/two_pass_pipeline_prof.py 2>/dev/null | cut -c1-170
done
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py 2>&1 >/dev/null | grep "fwdflat_kernel profile\|slowest\|fwdflat host" | tail -3
(timeout 1500 python -m pytest tests/test_zz_flat_gpu.py tests/test_dropin_gpu.py -m gpu -x -q -k "flat or 516 or second" 2>&1 | tail -3)
