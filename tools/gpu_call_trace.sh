set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r03u; mkdir -p $OUT
R=$PWD
cd /tmp
OP_PIPES=2 OP_STEPS=10 timeout 200 rocprofv3 --kernel-trace -f csv -d $OUT/ov2 -o ov -- python $R/tools/overlap_probe.py > $OUT/ov2.log 2>&1
cd $R
python tools/overlap_trace.py $OUT/ov2 > $OUT/overlap_trace.txt 2>&1
head -15 $OUT/overlap_trace.txt; grep "per step" $OUT/ov2.log
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
