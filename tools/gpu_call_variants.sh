#!/bin/bash
# A/B builds of the tree-search kernel (tools/build_variant.py) on the large-vocabulary leg: frames/s and the search stage's ms
#   usage: tools/gpu_call_variants.sh TAG name1 name2 ...   ("base" = the product library)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  lib=$PWD/pocketsphinx_amd/libpsgpu_$v.so; [ "$v" = base ] && lib=$PWD/pocketsphinx_amd/libpsgpu.so
  PSGPU_LIB_PATH=$lib timeout 300 python bench.py --workload large --steps 2 --no-cpu-baseline ${LV_ARGS:-} > "$OUT/$v.json" 2> "$OUT/$v.err"
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
    print("$v", j.get("value"), j.get("stage_ms", {}).get("search"), j.get("error"))
except Exception as e:
    print("$v: failed", e, open("$OUT/$v.err").read()[-300:])
PY
done
