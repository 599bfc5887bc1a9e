# one gpurun call: tools/overlap_probe.py without the profiler, then traced -> gpurun_out/$TAG
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/${TAG:-r03m}; mkdir -p $OUT
R=$PWD
export OP_STEPS=${OP_STEPS:-12}
(
 OP_PIPES=${OP_PIPES:-1,2} timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids) > $OUT/plain.log 2>&1
cat $OUT/plain.log
cd /tmp
OP_PIPES=2 timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/ov2 -o ov -- python $R/tools/overlap_probe.py > $OUT/ov2.log 2>&1
cd $R
python tools/overlap_trace.py $OUT/ov2 > $OUT/overlap_trace.txt 2>&1
head -16 $OUT/overlap_trace.txt
grep "per step\|completions" $OUT/ov2.log
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
