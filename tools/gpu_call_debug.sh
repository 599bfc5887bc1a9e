set -u
TAG=${1:-r02t}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests/test_ptm_gpu.py tests/test_fe_gpu.py tests/test_hmm_gpu.py tests/test_decode_pipeline_gpu.py tests/test_dropin_gpu.py -q -m gpu -k "not cmudict and not big" 2>&1 | tail -25 | tee $OUT/pytest.txt
(timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}') | tee $OUT/bench.txt
