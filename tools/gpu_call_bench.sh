#!/bin/bash
# one gpurun call: the bench exactly as the driver runs it (+ optional test files first)
set -u
TAG=${1:-b}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
if [ $# -gt 0 ]; then (timeout 2400 python -m pytest "$@" -m gpu -q -x 2>&1 | tail -25) | tee "$OUT/pytest.log"; fi
( time timeout 1500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ) 2>&1 | tail -3
python - "$OUT/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps({k: v for k, v in j.items() if k not in ("extra",)}, indent=None)[:6000])
print("extra keys:", list(j.get("extra", {}).keys()))
for k in ("device_decode_two_pass", "search_only_cmudict", "search_only_turtle"):
    print(k, json.dumps(j.get("extra", {}).get(k))[:800])
PY
tail -3 "$OUT/bench.err"
