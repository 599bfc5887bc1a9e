set -u
TAG=${1:-r03a}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}') | tee $OUT/bench.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc.log 2>&1)
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    vv = v[1:] if len(v) > 1 else v
    print("%-42s FETCH %.2f GB/launch (x2 corrected), %d launches" % (k, 2 * 1024 * sum(vv) / len(vv) / 1e9, len(v)))
PY
find $OUT -name "*.csv" -size +2M -delete
timeout 900 python -m pytest tests/test_ptm_gpu.py tests/test_decode_pipeline_gpu.py -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
