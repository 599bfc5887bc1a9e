#!/bin/bash
# round 6, call n: where the two-pass step's time goes outside the two search kernels (host timers + rocprofv3 API trace)
set -u
TAG=${1:-r6_n}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/two_pass_pipeline_prof.py > "$OUT/plain.json" 2> "$OUT/plain.err"
cat "$OUT/plain.json"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace --stats -d "$OUT/prof" -o tp -- python "$GRAFT_REPO_ROOT/tools/two_pass_pipeline_prof.py" > "$OUT/prof.json" 2> "$OUT/prof.err" )
cat "$OUT/prof.json"
for f in $(find "$OUT/prof" -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv" | head -4); do echo "== $f"; head -25 "$f" | cut -c1-160; done
find "$OUT/prof" -name "*trace.csv" -size +20M -delete
