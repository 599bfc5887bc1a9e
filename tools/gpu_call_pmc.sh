# search kernel alone: throughput + SQ counters (instruction mix, wait / active cycles)
set -u
TAG=${1:-r02g}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(SB_BATCHES=1,512 timeout 200 python tools/search_bench.py 2>&1 | tail -3) | tee $OUT/search_turtle.txt
(rocprofv3 -L 2>&1 | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ') > $OUT/avail_sq.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd /tmp && SB_BATCHES=512 SB_REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/search_bench.py > $OUT/pmc$i.log 2>&1)
  tail -2 $OUT/pmc$i.log
done
python - <<PY
import csv, glob, collections
for i in (1, 2, 3):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fwdtree_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("pass%d %-24s per-launch %.4g (%d launches)" % (i, k, v / max(n, 1), n))
PY
find $OUT -name "*.csv" -size +2M -delete
