#!/bin/bash
# round 6, call t: configs[2]'s shape (one 60 s utterance, 134,865 words, both passes): times, and both kernels' phase profiles
set -u
TAG=${1:-r6_t}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TP_TASK=big TP_SYNTH=60 TP_B=1 TP_CHECK_EVERY=1 timeout 600 python tools/two_pass_bench.py 2> "$OUT/plain.err" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ('seconds', 'first_pass_call_s', 'second_pass_call_s', 'status_nonzero')}, j.get('parity'))" | tee "$OUT/plain.txt"
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TP_TASK=big TP_SYNTH=60 TP_B=1 TP_CHECK_EVERY=0 timeout 600 python tools/two_pass_bench.py > /dev/null 2> "$OUT/prof.err"
grep -A30 "fwdflat_kernel profile" "$OUT/prof.err" | tail -31 | tee "$OUT/prof.txt"
grep "fwdflat host" "$OUT/prof.err" | tail -1
