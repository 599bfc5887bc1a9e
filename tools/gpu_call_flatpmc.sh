# second-pass kernel: SQ counters (instruction mix, waits, instruction fetch)
set -u
TAG=${1:-flatpmc}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(rocprofv3 -L 2>&1 | grep -o "SQC\?_[A-Z_0-9]*" | sort -u | tr '\n' ' ') > $OUT/avail_sq.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  i=$((i+1))
  (cd /tmp && TP_B=256 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/two_pass_bench.py > $OUT/pmc$i.log 2>&1)
  tail -1 $OUT/pmc$i.log | cut -c1-200
done
python - <<PY
import csv, glob, collections
for i in (1, 2, 3, 4):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("fwdflat_kernel", "fwdtree_kernel"):
                if kn in r["Kernel_Name"]:
                    a = acc[(kn, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("pass%d %-16s %-24s per-launch %.4g (%d launches)" % (i, k[0], k[1], v / max(n, 1), n))
PY
find $OUT -name "*.csv" -size +2M -delete
