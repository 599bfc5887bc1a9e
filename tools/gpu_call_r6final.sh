#!/bin/bash
# round 6: the round's measurement at one tree -- GPU suite (full), smoke, the bench line, rocprofv3 kernel stats and the PMC passes of the
# headline (tools/gpu_round.sh), then the large-vocabulary leg's kernel stats + PMC passes at 256 utterances
set -u
TAG=${1:-round6}
bash tools/gpu_round.sh $TAG - -
bash tools/gpu_call_lvpmc.sh ${TAG}_largevocab 256
# the second pass's phase profile at 512 x 30 s (profile build: tools/build_prof_lib.py) and its time against the utterances in flight
mkdir -p gpurun_out/profiles_${TAG}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ echo "# fwdflat_kernel<3, scoring> per-phase cycles per frame (PSGPU_FT_PROFILE build, tools/two_pass_pipeline_prof.py: the pipeline object's two passes, 512 x 30 s)";
  PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so TPP_STEPS=1 timeout 600 python tools/two_pass_pipeline_prof.py 2>&1 >/dev/null | grep -A34 "fwdflat_kernel profile" | tail -35 | grep -v "^fwdtree\|wavefront [123]\|evaluation over\|   0 top\|[0-9]\{9,\} cycles"; 
  echo "# the step's host timers, product build (ms; 3 steps)"; timeout 600 python tools/two_pass_pipeline_prof.py 2>/dev/null;
  echo "# time against the utterances in flight (tools/gpu_call_r6r.sh)"; bash tools/gpu_call_r6r.sh ${TAG}_r 2>/dev/null | grep "^== \|second_pass_call_ms\|fwdflat_kernel profile" | sed 's/.*"first_pass_wait_ms": \([0-9.]*\), "second_pass_call_ms": \([0-9.]*\).*/first pass \1 ms, second pass call \2 ms/'; } > gpurun_out/profiles_${TAG}/${TAG}_fwdflat_phase_profile.txt
cat gpurun_out/profiles_${TAG}/${TAG}_fwdflat_phase_profile.txt
