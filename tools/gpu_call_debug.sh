set -u
TAG=${1:-r03b}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_zz_flat_gpu.py -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
(TP_B=256 TP_NO_LISTS=1 timeout 300 python tools/two_pass_bench.py 2>&1 | tail -2) | tee $OUT/two_pass_scan.txt
(TP_B=256 timeout 300 python tools/two_pass_bench.py 2>&1 | tail -2) | tee $OUT/two_pass_lists.txt
(cd /tmp && TP_B=256 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/two_pass_bench.py > $OUT/stats.log 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-40s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -name "*.csv" -size +2M -delete
