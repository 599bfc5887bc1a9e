#!/bin/bash
# round 6, call v: the language model's cache of finished look-ups on / off (PSGPU_LM_CACHE): the large-vocabulary leg at 256 utterances,
# configs[2]'s shape (one 60 s utterance, both passes), the LM / large-vocabulary GPU tests
set -u
TAG=${1:-r6_v}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1800 python -m pytest tests/test_lm_gpu.py tests/test_search_gpu.py tests/test_zz_flat_gpu.py -m gpu -x -q 2>&1 | tail -3) | tee "$OUT/pytest.log"
for c in 1 0 1 0; do
  echo "== PSGPU_LM_CACHE=$c" | tee -a "$OUT/lv.txt"
  PSGPU_LM_CACHE=$c timeout 900 python bench.py --workload large --steps 2 --no-cpu-baseline --utts 256 --large-vocab-utts 256 2>> "$OUT/err.txt" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ('value', 'ms_per_step')}, j.get('roofline', {}).get('kernel_ms'))" | tee -a "$OUT/lv.txt"
  PSGPU_LM_CACHE=$c TP_TASK=big TP_SYNTH=60 TP_B=1 TP_CHECK_EVERY=1 timeout 600 python tools/two_pass_bench.py 2>> "$OUT/err.txt" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ('seconds', 'first_pass_call_s', 'second_pass_call_s')}, j.get('parity', {}).get('identical'))" | tee -a "$OUT/lv.txt"
done
