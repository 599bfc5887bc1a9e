set -u
TAG=${1:-r02o}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(SB_BATCHES=1,512 timeout 200 python tools/search_bench.py 2>&1 | tail -2) | tee $OUT/search_turtle.txt
(timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}') | tee $OUT/bench.txt
(PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "cycles/frame\|evaluation over" | tail -32) | tee $OUT/bench_prof.txt
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_search_layouts_gpu.py tests/test_decode_pipeline_gpu.py -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
