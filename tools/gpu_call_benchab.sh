#!/bin/bash
# headline line (no extras, no reference leg) for a list of libraries: tools/gpu_call_benchab.sh TAG variant...
set -u
TAG=${1:-bab}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/bench.txt"
  PSGPU_BENCH_NO_PCIE=1 PSGPU_LIB_PATH=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee -a "$OUT/bench.txt"
done
