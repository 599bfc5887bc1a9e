#!/bin/bash
# after the language-model look-up's flat loads were put right: LM / search / large-vocabulary parity, then the large-vocabulary leg and the headline
set -u
TAG=${1:-r6_m}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests/test_lm_set.py tests/test_lm_gpu.py tests/test_search_gpu.py tests/test_largevocab_gpu.py tests/test_zz_flat_gpu.py -m gpu -q -x 2>&1 | tail -6) > "$OUT/pytest.log"
cat "$OUT/pytest.log"
timeout 900 python bench.py --workload large --steps 2 --no-cpu-baseline > "$OUT/bench_large.json" 2> "$OUT/bench_large.err"
python - <<PY
import json
j = json.loads(open("$OUT/bench_large.json").read().strip().splitlines()[-1])
print("large:", {k: j.get(k) for k in ("value", "ms_per_step", "stage_ms", "error")}, j.get("roofline", {}).get("frac"))
PY
PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('headline:', j['value'], j['ms_per_step'], j['stage_ms'])
"
