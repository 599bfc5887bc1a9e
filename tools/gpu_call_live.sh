#!/bin/bash
# round 4: the resumable search / the live utterance -- parity tests of the three levels (kernel, pipeline, ps_searchfuncs binding),
# the search alone (its frame chain must not have slowed down), the headline's short line
set -u
TAG=${1:-live}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1200 python -m pytest tests/test_search_gpu.py tests/test_decode_pipeline_gpu.py tests/test_hmm_gpu.py "tests/test_dropin_gpu.py" ${PYTEST_MORE:-} -m gpu -q -x -k "${PYTEST_K:-resum or live or partial or session or phone_loop or vtable}" 2>&1 | tail -15) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
SB_BATCHES=512 SB_REPS=5 timeout 300 python tools/search_bench.py 2>&1 | grep "B=" | tee "$OUT/search.txt"
if [ "${BENCH:-1}" = "1" ]; then
  PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee "$OUT/bench.txt"
fi
