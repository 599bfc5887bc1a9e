export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/$1
for b in 256 512 1024 2048; do
TP_B=$b timeout 600 python tools/two_pass_bench.py > gpurun_out/$1/two_pass_$b.json 2> gpurun_out/$1/two_pass_err_$b.txt
python -c "
import json; j=json.loads(open('gpurun_out/$1/two_pass_$b.json').read().strip().splitlines()[-1]); print($b, j['first_pass_call_s'], j['second_pass_call_s'], j['parity']['identical'], j['parity']['checked'], j['frames_per_s'])"
done
