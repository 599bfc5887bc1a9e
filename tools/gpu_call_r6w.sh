#!/bin/bash
# round 6, call w: the runtime's busy thread (58 ms of CPU per 76 ms step, profiles/round5_host_cpu.txt): which setting moves it
set -u
TAG=${1:-r6_w}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
env | grep -i "^HSA\|^HIP\|^AMD\|^ROC\|^GPU_" | tee "$OUT/env.txt"
run() { echo "== $*" | tee -a "$OUT/out.txt"; env "$@" timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 2>> "$OUT/err.txt" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); h = j.get('host', {})
print(j['ms_per_step'], h.get('cpu_ms_per_step_per_rank'), h.get('busiest_threads_ms_per_step'), h.get('main_thread_ms_per_step'))" | tee -a "$OUT/out.txt"; }
run X=1
run HSA_ENABLE_INTERRUPT=0
run HSA_ENABLE_INTERRUPT=1
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=2
run HIP_LAUNCH_BLOCKING=0 ROC_ACTIVE_WAIT_TIMEOUT=0
run PSGPU_BENCH_PIPES=1
