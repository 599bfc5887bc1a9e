#!/bin/bash
# round 5, call a: the whole GPU suite (not -x: every failure at once) after the binding fix + ASan cases
set -u
OUT=$PWD/gpurun_out/r5_a; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -120) > "$OUT/pytest_gpu.log"
tail -60 "$OUT/pytest_gpu.log"
