#!/bin/bash
# rocprofv3 kernel durations of the attention micro-benchmark (tools/bench_mha.py), one process per shape.
# usage: tools/prof_mha.sh <tag>  -> gpurun_out/mha_<tag>.txt   (DTYPE=bf16|f16 for the 16-bit contractions)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-x}
shapes=${SHAPES:-"1024x1024 80x80 80x1024 1024x80 1024x132 256x256 256x80 256x132 256x1024"}
mkdir -p gpurun_out
: > gpurun_out/mha_${tag}.txt
for sh in $shapes; do
  out=/tmp/prof_mha_$sh
  rm -rf $out
  ONLY=$sh ITERS=${ITERS:-10} rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/bench_mha.py > /tmp/prof_mha.log 2>&1
  t=$(find $out -name '*kernel_trace.csv' | head -1)
  echo "== $sh" >> gpurun_out/mha_${tag}.txt
  python tools/mha_trace_summary.py "$t" >> gpurun_out/mha_${tag}.txt
done
