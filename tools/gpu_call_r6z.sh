#!/bin/bash
# round 6, call z: the headline with a variant library beside the product (twice each): ms per step, the search kernel's time beside / alone
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
lib() { [ "$1" = "default" ] && echo $PWD/pocketsphinx_amd/libpsgpu.so || echo $PWD/pocketsphinx_amd/libpsgpu_$1.so; }
for v in $1 $1; do
  echo "== $v"
  PSGPU_LIB_PATH=$(lib $v) timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print(j['value'], j['ms_per_step'], 'search beside', r.get('kernel_ms'), 'alone', r.get('kernel_ms_alone'), 'stages', j.get('stage_ms'))"
done
