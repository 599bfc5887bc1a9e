#!/bin/bash
# tools/gpu_call_largevocab.sh -- one gpurun call for the large-vocabulary leg (134,865 words): its parity tests, the leg as a
# bench line (reference cpu_baseline + per-utterance parity), the phase profile of fwdtree_kernel<3, 1024, false, false>
# (PSGPU_FT_PROFILE build), rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, never with
# a trace domain other than --kernel-trace).  Everything lands under gpurun_out/<tag>/.
#   usage: tools/gpu_call_largevocab.sh TAG [skip-tests] [full-bench]
set -u
TAG=${1:-lv}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
nproc > "$OUT/host.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/host.txt"

if [ "${2:-}" != "skip-tests" ]; then
  (timeout 1500 python -m pytest tests/test_largevocab_gpu.py -q -x 2>&1 | tail -30) > "$OUT/pytest_largevocab.log"
  cat "$OUT/pytest_largevocab.log"
fi
timeout 900 python bench.py --workload large --steps 2 > "$OUT/bench_large.json" 2> "$OUT/bench_large.err"
tail -c 5000 "$OUT/bench_large.json"; tail -5 "$OUT/bench_large.err"

# phase profile (cycles per frame and phase, work-item 0) at B = 64 and B = 1
if [ -f pocketsphinx_amd/libpsgpu_prof.so ]; then
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline \
      --utts 64 --large-vocab-utts 64 > "$OUT/prof64.json" 2> "$OUT/phase_profile_b64.txt"
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline \
      --utts 1 --large-vocab-utts 1 > "$OUT/prof1.json" 2> "$OUT/phase_profile_b1.txt"
  grep -v "^$" "$OUT/phase_profile_b64.txt" | tail -45
fi

BENCH="python $PWD/bench.py --workload large --steps 1 --no-cpu-baseline --utts 64 --large-vocab-utts 64"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -T -f csv -d "$OUT/pmc_sq" -o sq -- $BENCH > "$OUT/pmc_sq.log" 2>&1
cd - > /dev/null
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
python tools/prof_collect.py "$OUT" "$TAG" 64 30.0 2>&1 | tail -50
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fwdtree_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("fwdtree_kernel %-24s per-launch %.4g (%d launches)" % (k, v / max(n, 1), n))
PY
if [ "${3:-}" = "full-bench" ]; then
  timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
