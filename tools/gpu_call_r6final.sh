#!/bin/bash
# round 6: the round's measurement at one tree -- GPU suite (full), smoke, the bench line, rocprofv3 kernel stats and the PMC passes of the
# headline (tools/gpu_round.sh), then the large-vocabulary leg's kernel stats + PMC passes at 256 utterances
set -u
TAG=${1:-round6}
bash tools/gpu_round.sh $TAG - -
bash tools/gpu_call_lvpmc.sh ${TAG}_largevocab 256
