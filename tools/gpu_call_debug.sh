set -u
TAG=${1:-r02w}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "1 1" "2 1" "4 1" "1 4" "1 2"; do
  set -- $cfg
  echo "=== PSGPU_LANE_FPL=$1 PSGPU_SEN_FR=$2"
  (PSGPU_LANE_FPL=$1 PSGPU_SEN_FR=$2 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -c 1500 | grep -o '"value": [0-9.]*\|"stage_ms": {[^}]*}')
done 2>&1 | tee $OUT/variants.txt
timeout 1700 python -m pytest tests/test_decode_pipeline_gpu.py tests/test_ptm_gpu.py -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
(cd /tmp && PSGPU_LANE_FPL=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats.log 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-40s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -name "*.csv" -size +2M -delete
