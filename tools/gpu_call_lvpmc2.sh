#!/bin/bash
# one PMC pass of the large-vocabulary leg's search kernel with the named counters:  TAG N counter...
set -u
TAG=$1; N=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH="python $PWD/bench.py --workload large --steps 1 --no-cpu-baseline --utts $N --large-vocab-utts $N"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc "$@" -T -f csv -d "$OUT/pmc" -o c -- $BENCH > "$OUT/pmc.log" 2>&1
cd - > /dev/null
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
tail -3 "$OUT/pmc.log"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fwdtree_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("fwdtree_kernel %-28s per-launch %.4g (%d launches)" % (k, v / max(n, 1), n))
PY
