export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$1
PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TP_B=256 timeout 600 python tools/two_pass_bench.py > gpurun_out/$1/two_pass_prof.json 2> gpurun_out/$1/two_pass_prof.txt
grep -A9 "fwdflat_kernel profile" gpurun_out/$1/two_pass_prof.txt | tail -10
TP_B=256 timeout 600 python tools/two_pass_bench.py > gpurun_out/$1/two_pass.json 2> gpurun_out/$1/two_pass_err.txt
python -c "
import json; j=json.loads(open('gpurun_out/$1/two_pass.json').read().strip().splitlines()[-1]); print(j['first_pass_call_s'], j['second_pass_call_s'], j['parity'], j['utt0_second_pass_table_is_reference'])"
timeout 900 python -m pytest tests/test_zz_flat_gpu.py -x -q 2>&1 | tail -3
