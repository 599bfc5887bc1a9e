#!/bin/bash
# round 5, call b: the compact-channel slab layouts on the device -- parity tests of the search (all layouts) and of the
# large-vocabulary pipeline, the large-vocabulary leg as a bench line, its phase profile (PSGPU_FT_PROFILE build)
set -u
TAG=${1:-r5_b}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_search_layouts_gpu.py tests/test_largevocab_gpu.py tests/test_decode_pipeline_gpu.py -m gpu -q 2>&1 | tail -30) > "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
timeout 900 python bench.py --workload large --steps 2 > "$OUT/bench_large.json" 2> "$OUT/bench_large.err"
python - <<PY
import json
try:
    j = json.loads(open("$OUT/bench_large.json").read().strip().splitlines()[-1])
    print({k: j.get(k) for k in ("value", "ms_per_step", "stage_ms", "parity", "error")}, j.get("roofline", {}).get("frac"), j.get("config", {}).get("search_slab_bytes_per_utterance"))
except Exception as e:
    print("bench_large:", e)
PY
tail -3 "$OUT/bench_large.err"
if [ -f pocketsphinx_amd/libpsgpu_prof.so ]; then
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline \
      --utts 64 --large-vocab-utts 64 > "$OUT/prof64.json" 2> "$OUT/phase_profile_b64.txt"
  grep -v "^$" "$OUT/phase_profile_b64.txt" | tail -45
fi
