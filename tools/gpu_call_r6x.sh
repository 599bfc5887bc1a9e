#!/bin/bash
# round 6, call x: the live streams' step with the wait spinning (default) and polling (PSGPU_POLL_WAIT_US): step time and the process's CPU
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 20 50; do
  echo "== PSGPU_POLL_WAIT_US=$v"
  PSGPU_POLL_WAIT_US=$v timeout 300 python - > /tmp/sb.json 2> /tmp/sb.err <<'P'
import resource, runpy, sys, time
t0 = time.time()
try:
    runpy.run_path("tools/streams_bench.py", run_name="__main__")
finally:
    r = resource.getrusage(resource.RUSAGE_SELF)
    sys.stderr.write("process CPU: user %.2f s, sys %.2f s, wall %.2f s\n" % (r.ru_utime, r.ru_stime, time.time() - t0))
P
  python - <<'P'
import json
j = json.loads(open('/tmp/sb.json').read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("ms_per_step", "step_ms_median", "step_ms_p99", "steps", "final_hypotheses_equal_the_one_call_decode")})
P
  tail -1 /tmp/sb.err
done
