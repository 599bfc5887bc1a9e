#!/bin/bash
# round 6, call f: the senone kernel's direct-store form (no staged row in LDS) -- parity, then the headline with it and with the staged form
set -u
TAG=${1:-r6_f}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_ptm_gpu.py tests/test_decode_pipeline_gpu.py tests/test_scorers_pipeline_gpu.py tests/test_random_models_gpu.py -m gpu -q -x 2>&1 | tail -8) > "$OUT/pytest.log"
cat "$OUT/pytest.log"
for V in 0 1 0 1; do
  echo "== PSGPU_SENONE_STAGED=$V" | tee -a "$OUT/bench.txt"
  PSGPU_SENONE_STAGED=$V PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'], j['roofline'].get('scorer', {}).get('kernel_ms'), j.get('parity'))
" | tee -a "$OUT/bench.txt"
done
