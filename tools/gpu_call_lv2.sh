#!/bin/bash
# one gpurun call while working on the slab-layout search kernel: its parity tests, the large-vocabulary bench line, the phase profile
set -u
TAG=${1:-lv2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests/test_largevocab_gpu.py tests/test_zz_search_layouts_gpu.py tests/test_lm_gpu.py -q -x 2>&1 | tail -15) > "$OUT/pytest.log"
cat "$OUT/pytest.log"
timeout 900 python bench.py --workload large --steps 2 > "$OUT/bench_large.json" 2> "$OUT/bench_large.err"
tail -c 2500 "$OUT/bench_large.json"; tail -3 "$OUT/bench_large.err"
if [ -f pocketsphinx_amd/libpsgpu_prof.so ]; then
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline \
      --utts 64 --large-vocab-utts 64 > "$OUT/prof64.json" 2> "$OUT/phase_profile_b64.txt"
  grep -v "^$" "$OUT/phase_profile_b64.txt" | tail -36
fi
if [ "${2:-}" = "medium" ]; then
  SB_CASE=medium_goforward SB_BATCHES=512 timeout 300 python tools/search_bench.py 2>&1 | tail -2
  PSGPU_FWDTREE_LAYOUT=slab SB_BATCHES=512 timeout 300 python tools/search_bench.py 2>&1 | tail -2
fi
