set -u
OUT=$PWD/gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(DBG_REPS=16 timeout 400 python tools/debug_long_utt.py 2>&1 | tail -30) > $OUT/debug_long_1.txt; cat $OUT/debug_long_1.txt
(DBG_REPS=10 DBG_COPIES=3 timeout 400 python tools/debug_long_utt.py 2>&1 | tail -24) > $OUT/debug_long_3.txt; cat $OUT/debug_long_3.txt
