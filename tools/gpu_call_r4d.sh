#!/bin/bash
# round 4, call d: the search kernel after a step of work on its frame chain -- parity tests, the search alone, the phase profile
set -u
TAG=${1:-r4d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_search_gpu.py tests/test_decode_pipeline_gpu.py tests/test_zz_search_layouts_gpu.py ${PYTEST_MORE:-} -m gpu -q -x 2>&1 | tail -8) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
SB_BATCHES=512,1024 SB_REPS=5 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee "$OUT/search.txt"
SB_CASE=medium_goforward SB_BATCHES=512 SB_REPS=3 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee -a "$OUT/search.txt"
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so SB_BATCHES=512 SB_REPS=1 timeout 300 python tools/search_bench.py > "$OUT/prof.txt" 2>&1
tail -42 "$OUT/prof.txt"
if [ "${BENCH:-0}" = "1" ]; then
  PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee "$OUT/bench.txt"
fi
