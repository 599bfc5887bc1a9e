#!/bin/bash
# the whole GPU suite, then the headline line alone: tools/gpu_call_full.sh TAG
set -u
TAG=${1:-full}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee "$OUT/bench.txt"
tail -3 "$OUT/bench.err"
