#!/bin/bash
# SQ counters of the scorer's kernels in the headline workload: how much of the senone kernel's time is LDS (bank conflicts of its
# log-add table look-ups) and how much VALU
set -u
TAG=${1:-senpmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
BENCH="python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
cd /tmp
PSGPU_BENCH_PIPES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -f csv -d $OUT/a -o a -- $BENCH > $OUT/a.log 2>&1
PSGPU_BENCH_PIPES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -f csv -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no counters", open("$OUT/%s.log" % tag).read()[-400:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in acc:
        if any(x in k for x in ("senone", "lane", "fwdtree")):
            print(tag, k, {c: "%.3g" % (v / max(1, cnt[(k, c)])) for c, v in acc[k].items()})
PY
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
