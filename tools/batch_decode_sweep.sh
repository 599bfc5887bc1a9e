#!/bin/bash
# full-decode throughput (reference search on host threads, GMM scoring on the MI355X)
# next to the reference's CPU path on the same host; results -> gpurun_out/<tag>_batch_decode.jsonl
TAG=${1:-r01}
R=oracle/_ref
OUT=gpurun_out/${TAG}_batch_decode.jsonl
mkdir -p gpurun_out; : > $OUT
echo "host cores: $(nproc)" | tee -a $OUT
for mode in cpu gpu; do
  for th in 1 2 4 8 16 32; do
    [ $th -gt $(nproc) ] && continue
    n=$((th * 12))
    timeout 600 $R/psgpu_batch_decode $R/model/en-us $R/data/turtle.lm.bin $R/data/turtle.dic $R/data/goforward.raw $n $th $mode fwdflat no bestpath no | tee -a $OUT
  done
done
for mode in cpu gpu; do
  th=$(nproc); [ $th -gt 16 ] && th=16
  timeout 600 $R/psgpu_batch_decode $R/model/en-us $R/data/turtle.lm.bin $R/data/turtle.dic $R/data/goforward.raw $((th*8)) $th $mode | tee -a $OUT
done
