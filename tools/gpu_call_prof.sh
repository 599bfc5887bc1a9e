export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$1
N=${2:-64}
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so timeout 900 python bench.py --workload large --steps 1 --no-cpu-baseline --utts $N --large-vocab-utts $N > gpurun_out/$1/prof$N.json 2> gpurun_out/$1/phase_profile_b$N.txt
grep -v "^$" gpurun_out/$1/phase_profile_b$N.txt | tail -38
