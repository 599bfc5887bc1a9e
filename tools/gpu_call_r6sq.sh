#!/bin/bash
# SQ counters of the two search kernels -> gpurun_out/<tag>/sq_issue.json (committed as profiles/<round>_sq_issue.json; bench.py reads
# the newest for roofline.issue_frac / lv_issue_frac): how busy a resident wavefront's instruction stream is while the kernel holds its
# compute units = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, and what it waits for.  One PMC pass per workload (--kernel-trace only).
set -u
TAG=${1:-r6_sq}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
cd /tmp
PSGPU_BENCH_PIPES=1 timeout 600 rocprofv3 --kernel-trace --pmc $CNT -T -f csv -d "$OUT/head" -o sq -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > "$OUT/head.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc $CNT -T -f csv -d "$OUT/lv" -o sq -- python $R/bench.py --workload large --steps 1 --no-cpu-baseline > "$OUT/lv.log" 2>&1
cd $R
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
python - <<PY
import csv, glob, collections, json
out = {}
for key, sub, what in (("fwdtree_kernel_headline", "head", "bench.py --no-extras, 512 x 30 s, one pipeline object (the kernel alone)"),
                       ("fwdtree_kernel_large_vocab", "lv", "bench.py --workload large, 256 x 30 s")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fwdtree_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    c = {k: v / max(n, 1) for k, (v, n) in acc.items()}
    if not c.get("SQ_WAVE_CYCLES"):
        print(key, "no counters:", open("$OUT/%s.log" % sub).read()[-300:]); continue
    out[key] = {"counters_per_launch": {k: float("%.4g" % v) for k, v in sorted(c.items())}, "launches": int(acc["SQ_WAVE_CYCLES"][1]), "workload": what,
                "issue_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                "wait_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                "valu_active_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                "lds_active_frac": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                "what": "issue_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES: the share of a resident wavefront's cycles in which it has an instruction in flight; "
                        "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share it is parked waiting (memory, LDS, barriers)"}
    print(key, {k: out[key][k] for k in ("issue_frac", "wait_frac", "valu_active_frac", "lds_active_frac", "launches")})
json.dump(out, open("$OUT/sq_issue.json", "w"), indent=1, sort_keys=True)
PY
