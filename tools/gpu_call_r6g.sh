#!/bin/bash
# round 6, call g: the second pass's exit queue -- 96 / 128 / 160 (default) / 256 entries: static LDS 55 / 59 / 63 / 74 KB + 20.5 KB dynamic
set -u
TAG=${1:-r6_g}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in default ffx128 ffx256 default ffx128 ffx256 ffx128 default; do
  [ "$v" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$v.so
  echo "== $v" | tee -a "$OUT/two_pass.txt"
  PSGPU_LIB_PATH=$L TP_SYNTH=30 TP_B=512 TP_CHECK_EVERY=128 timeout 600 python tools/two_pass_bench.py 2>>"$OUT/err.txt" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ('seconds', 'first_pass_call_s', 'second_pass_call_s', 'status_nonzero')}, j['parity'].get('identical'), j['parity'].get('checked'))
" | tee -a "$OUT/two_pass.txt"
done
