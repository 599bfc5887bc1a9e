export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$1
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TP_B=256 timeout 600 python tools/two_pass_bench.py > gpurun_out/$1/two_pass_prof.json 2> gpurun_out/$1/two_pass_prof.txt
grep "fwdflat host" gpurun_out/$1/two_pass_prof.txt
grep -A14 "fwdflat_kernel profile" gpurun_out/$1/two_pass_prof.txt | tail -15
