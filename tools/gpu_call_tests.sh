#!/bin/bash
# one gpurun call: the given GPU test files (default: all), output tail into gpurun_out/<tag>/pytest.log
set -u
TAG=${1:-t}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 2700 python -m pytest ${@:-tests} -m gpu -q -x 2>&1 | tail -40) > "$OUT/pytest.log"
cat "$OUT/pytest.log"
