#!/bin/bash
# the bench line as the driver runs it (all legs and extras), timed: tools/gpu_call_bench_full.sh TAG
set -u
TAG=${1:-bf}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench.py wall: $(( $(date +%s) - t0 )) s" | tee "$OUT/wall.txt"
python - "$OUT/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j.get("roofline", {}).get("frac"), j.get("parity"))
for k in ("decode_ms_scorer", "decode_two_pass", "decode_large_vocab"):
    v = j.get(k, {}); print(k, {q: v.get(q) for q in ("frames_per_s", "value", "ms_per_step", "first_pass_ms", "second_pass_ms", "parity", "error")})
for k, v in j.get("extra", {}).items():
    print(k, json.dumps(v)[:600])
PY
tail -5 "$OUT/bench.err"
