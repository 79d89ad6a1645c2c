#!/bin/bash
# csrc/mha4.hip (keys per wave) against mha2.hip's key-split forward: rocprofv3 kernel durations of the short-query / long-key shapes.
# usage: tools/bench_mha4.sh -> gpurun_out/mha4.txt
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/mha4.txt
: > $out
for cfg in "EDA_MHA4=0" "EDA_MHA4=1" "EDA_MHA4=2"; do
  echo "#### $cfg" >> $out
  env $cfg SHAPES="80x1024 130x1024 256x1024" bash tools/prof_mha.sh m4_tmp > /dev/null 2>&1
  grep -E "^==|fwd_kernel" gpurun_out/mha_m4_tmp.txt >> $out
done
rm -f gpurun_out/mha_m4_tmp.txt
