#!/bin/bash
# round 4, call c: the senone kernel's biased form (parity tests, headline bench), then the headline with searches of the two
# pipeline objects allowed to be resident together on builds of the search kernel that leave room for a third workgroup per CU
set -u
TAG=${1:-r4c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_ptm_gpu.py tests/test_ptm_frame_gpu.py tests/test_decode_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -8) > "$OUT/pytest.log"; cat "$OUT/pytest.log"
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3"
run() {  # name lib pipes overlap
  [ "$2" = "default" ] && L=$PWD/pocketsphinx_amd/libpsgpu.so || L=$PWD/pocketsphinx_amd/libpsgpu_$2.so
  echo "== $1" | tee -a "$OUT/bench.txt"
  PSGPU_BENCH_NO_PCIE=1 PSGPU_LIB_PATH=$L PSGPU_BENCH_PIPES=$3 PSGPU_DECODE_SEARCH_OVERLAP=$4 PSGPU_SENONE_SAD=${5:-1} timeout 300 $B 2>>"$OUT/bench.err" | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j['value'], j['ms_per_step'], j['stage_ms'], j['stage_ms_one_step_alone'])
" | tee -a "$OUT/bench.txt"
}
run default_p2 default 2 0
run default_p2_sad0 default 2 0 0
run rowsdev_p2_ov rowsdev 2 1
run rowsdev_p3_ov rowsdev 3 1
run rowsdev128_p2_ov rowsdev128 2 1
run rowsdev128_p3_ov rowsdev128 3 1
run rowsdev128_p2 rowsdev128 2 0
tail -5 "$OUT/bench.err"
