set -u
TAG=${1:-r02k}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "" _os; do
  echo "=== variant libpsgpu$v"
  (PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu$v.so SB_BATCHES=1,512 timeout 200 python tools/search_bench.py 2>&1 | tail -2) | tee $OUT/search_turtle$v.txt
done
(timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}') | tee $OUT/bench.txt
(PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "cycles/frame" | tail -28) | tee $OUT/bench_prof.txt
rocprofv3 -L 2>&1 | grep -o "\b[A-Z][A-Za-z]*_[A-Z_0-9a-z]*\b" | sort -u | tr '\n' ' ' > $OUT/avail_all.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  (cd /tmp && SB_BATCHES=512 SB_REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/search_bench.py > $OUT/pmc$i.log 2>&1)
  tail -2 $OUT/pmc$i.log
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fwdtree_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("pass%d %-24s per-launch %.4g (%d launches)" % (i, k, v / max(n, 1), n))
PY
find $OUT -name "*.csv" -size +2M -delete
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_search_layouts_gpu.py tests/test_decode_pipeline_gpu.py -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
