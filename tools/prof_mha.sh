#!/bin/bash
# rocprofv3 kernel durations of the attention micro-benchmark (tools/bench_mha.py), one process per shape.
# usage: tools/prof_mha.sh <tag> [impls: "2 1"]  -> gpurun_out/mha_<tag>_{new,old}.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-x}
impls=${2:-"2 1"}
shapes=${SHAPES:-"1024x1024 80x80 80x1024 1024x80 1024x132 256x256 256x80 256x132 256x1024"}
for impl in $impls; do
  name=new; [ $impl = 1 ] && name=old
  : > gpurun_out/mha_${tag}_${name}.txt
  for sh in $shapes; do
    out=/tmp/prof_${name}_$sh
    rm -rf $out
    ONLY=$sh EDA_MHA_IMPL=$impl ITERS=${ITERS:-10} rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/bench_mha.py > /tmp/prof_$name.log 2>&1
    t=$(find $out -name '*kernel_trace.csv' | head -1)
    echo "== $sh" >> gpurun_out/mha_${tag}_${name}.txt
    python tools/mha_trace_summary.py "$t" >> gpurun_out/mha_${tag}_${name}.txt
  done
done
