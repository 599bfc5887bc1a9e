set -u
TAG=${1:-r03c}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 600 python tools/overlap_probe.py 2>&1 | tail -6) | tee $OUT/overlap.txt
(GPU_MAX_HW_QUEUES=8 timeout 600 python tools/overlap_probe.py 2>&1 | tail -6) | tee $OUT/overlap_q8.txt
