set -u
OUT=$PWD/gpurun_out/r02f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "" _plain _direct; do
  echo "=== variant libpsgpu$v"
  (PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu$v.so SB_BATCHES=1,512 timeout 200 python tools/search_bench.py 2>&1 | tail -2) | tee $OUT/search_turtle$v.txt
  (PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu$v.so PSGPU_BENCH_NO_OVERLAP=1 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}') | tee $OUT/bench$v.txt
done
(PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof_direct.so SB_BATCHES=512 SB_REPS=1 timeout 200 python tools/search_bench.py 2>&1 | tail -21) | tee $OUT/search_turtle_prof_direct.txt
