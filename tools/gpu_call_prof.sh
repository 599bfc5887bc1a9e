export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$1
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 600 python bench.py --workload large --steps 1 --no-cpu-baseline --utts 64 --large-vocab-utts 64 > gpurun_out/$1/prof64.json 2> gpurun_out/$1/phase_profile_b64.txt
grep -v "^$" gpurun_out/$1/phase_profile_b64.txt | tail -40
