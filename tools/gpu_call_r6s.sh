#!/bin/bash
# round 6, call s: the large-vocabulary leg, variants side by side (256 utterances; parity on a sample), and the phase profile at 64
# usage: gpu_call_r6s.sh TAG "lib1 lib2 ..."  (names without libpsgpu_; "default" = the product)
set -u
TAG=${1:-r6_s}; PL=${2:-default}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
lib() { [ "$1" = "default" ] && echo $PWD/pocketsphinx_amd/libpsgpu.so || echo $PWD/pocketsphinx_amd/libpsgpu_$1.so; }
for v in $PL $PL; do
  echo "== $v" | tee -a "$OUT/lv.txt"
  PSGPU_LIB_PATH=$(lib $v) timeout 900 python bench.py --workload large --steps 2 --no-cpu-baseline --utts 256 --large-vocab-utts 256 2>> "$OUT/err.txt" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ('value', 'ms_per_step', 'status_nonzero')}, j.get('roofline', {}).get('kernel_ms'))" | tee -a "$OUT/lv.txt"
done
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 900 python bench.py --workload large --steps 1 --no-cpu-baseline --utts 64 --large-vocab-utts 64 > /dev/null 2> "$OUT/prof.err"
grep -A45 "fwdtree_kernel profile" "$OUT/prof.err" | tail -46 | tee "$OUT/prof.txt"
