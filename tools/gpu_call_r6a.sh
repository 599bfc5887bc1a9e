#!/bin/bash
# round 6, call a: the round's first changes on the device -- bench.py's launcher contract, streams in the slab layout, the large-vocabulary
# capacities; then the phase profiles (PSGPU_FT_PROFILE build) of the headline's LDS layout and of the large-vocabulary leg at this tree
set -u
TAG=${1:-r6_a}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests/test_bench_launch.py tests/test_decode_pipeline_gpu.py tests/test_largevocab_gpu.py -m gpu -q 2>&1 | tail -30) > "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
if [ -f pocketsphinx_amd/libpsgpu_prof.so ]; then
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so PSGPU_BENCH_PIPES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras \
      > "$OUT/prof_headline.json" 2> "$OUT/phase_profile_headline.txt"
  grep -v "^$" "$OUT/phase_profile_headline.txt" | tail -60
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline \
      --utts 64 --large-vocab-utts 64 > "$OUT/prof64.json" 2> "$OUT/phase_profile_b64.txt"
  grep -v "^$" "$OUT/phase_profile_b64.txt" | tail -60
fi
