#!/bin/bash
# kernel trace of a live decode through the ps_searchfuncs binding (30 s synthetic utterance, 250 ms pieces, a read-out each)
set -u
TAG=${1:-livetrace}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
python -c "
import sys; sys.path.insert(0, '$R')
from pocketsphinx_amd import synth
synth.utterance(5, 30.0).tofile('/tmp/long.raw')
"
REF=$R/oracle/_ref
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tr -o live -- $REF/dropin_decode $REF/model/en-us $REF/data/turtle.lm.bin $REF/data/turtle.dic /tmp/long.raw 2 psgpu_device_vtable yes chunked ${CHUNK:-4000} fwdflat no bestpath no > $OUT/run.json 2> $OUT/run.err
cd $R
python - <<PY
import csv, glob
f = glob.glob('$OUT/tr/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print('%-60s calls %6s avg_us %10.1f total_ms %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
python -c "
import json; j = json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('ok','decode_s_cpu','decode_s_gpu','partial_results','live_steps','live_frames_searched','live_utt_frames')})"
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
