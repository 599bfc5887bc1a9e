#!/bin/bash
# kernel trace of the live_streams bench (512 streams, 100 ms a step): where a step's time goes
set -u
TAG=${1:-streamtrace}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
cd /tmp
LS_SEC=${LS_SEC:-10} timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tr -o st -- python $R/tools/streams_bench.py > $OUT/run.json 2> $OUT/run.err
cd $R
python - <<PY
import csv, glob
f = glob.glob('$OUT/tr/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print('%-60s calls %6s avg_us %10.1f total_ms %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
tail -1 $OUT/run.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print({k:j[k] for k in ('value','ms_per_step','step_ms_median','steps')})"
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
