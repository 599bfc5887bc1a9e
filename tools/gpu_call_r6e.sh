#!/bin/bash
# round 6, call e: the scorer in ranges (top-N pass of range r beside the senone pass of range r - 1): parity tests of the scorer and the
# pipeline, then the headline with 1 / 4 / 6 / 8 ranges
set -u
TAG=${1:-r6_e}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(PSGPU_PTM_RANGES=3 timeout 900 python -m pytest tests/test_ptm_gpu.py tests/test_decode_pipeline_gpu.py tests/test_scorers_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -8) > "$OUT/pytest_ranges3.log"
cat "$OUT/pytest_ranges3.log"
for R in 1 4 6 8; do
  echo "== ranges $R" | tee -a "$OUT/bench.txt"
  PSGPU_PTM_RANGES=$R PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'], j['roofline'].get('scorer', {}).get('kernel_ms'))
" | tee -a "$OUT/bench.txt"
done
tail -3 "$OUT/bench.err"
