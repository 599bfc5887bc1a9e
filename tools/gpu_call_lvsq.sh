#!/bin/bash
# SQ counters of the large-vocabulary leg's search kernel (one PMC pass, --kernel-trace only)
set -u
TAG=${1:-lvsq}; N=${2:-64}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH="python $PWD/bench.py --workload large --steps 1 --no-cpu-baseline --utts $N --large-vocab-utts $N"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -T -f csv -d "$OUT/pmc_sq" -o sq -- $BENCH > "$OUT/pmc_sq.log" 2>&1
cd - > /dev/null
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
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
