mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
for g in 0 1; do
PSGPU_SENONE_GENERIC=$g PSGPU_CHAIN_OCC=7 PSGPU_CHUNK=32 timeout 300 python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic=$g', d['value'], d['kernels_ms'])"
done
