#!/bin/bash
# both passes on the device (tools/two_pass_bench.py), A/B of libraries: tools/gpu_call_flatab.sh TAG variant...
set -u
TAG=${1:-fab}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  for cfg in "TP_B=256" "TP_B=512 TP_SYNTH=30"; do
    echo "== $v $cfg" | tee -a "$OUT/flat.txt"
    env PSGPU_LIB_PATH=$L $cfg timeout 400 python tools/two_pass_bench.py 2>>"$OUT/flat.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print({k: j.get(k) for k in ('frames_per_s','first_pass_call_s','second_pass_call_s','status_nonzero','parity')})
" | tee -a "$OUT/flat.txt"
  done
done
tail -3 "$OUT/flat.err"
