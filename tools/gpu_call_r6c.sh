#!/bin/bash
# round 6, call c: where the second pass's time goes at 512 x 30 s (PSGPU_FT_PROFILE build prints the host / kernel split and the phases);
# the launcher tests with their durations
set -u
TAG=${1:-r6_c}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_bench_launch.py -m gpu -q --durations=5 2>&1 | tail -15) > "$OUT/pytest.log"
cat "$OUT/pytest.log"
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TP_SYNTH=30 TP_B=512 TP_CHECK_EVERY=64 timeout 900 python tools/two_pass_bench.py > "$OUT/two_pass_512x30.json" 2> "$OUT/two_pass_512x30.err"
grep -v "^$" "$OUT/two_pass_512x30.err" | grep -i "fwdflat\|cycles/frame" | tail -40
python - <<PY
import json
j = json.loads(open("$OUT/two_pass_512x30.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("seconds", "first_pass_call_s", "second_pass_call_s", "parity", "status_nonzero")})
PY
