#!/bin/bash
# rocprofv3 kernel durations of the tiled row-GEMM shapes of the step (own kernel + the library's, same process).
# usage: tools/prof_gemm_shapes.sh <tag>  -> gpurun_out/gemm_shapes_<tag>.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-x}
shapes=${SHAPES:-"2048x288x288 640x3072x768 8192x288x288 640x768x3072 640x768x2304 640x768x768 640x288x288 2048x288x256 8192x288x576 8192x576x288 640x3456x288 1056x3456x288 2048x288x864"}
mkdir -p gpurun_out
out_txt=gpurun_out/gemm_shapes_${tag}.txt
echo "# rocprofv3 --kernel-trace, ${ITERS:-20} launches per kernel (first two dropped), fp32; EDA_GEMM_SPLITK=${EDA_GEMM_SPLITK:-default}" > $out_txt
for sh in $shapes; do
  out=/tmp/prof_gemm_$sh
  rm -rf $out
  SHAPES=$sh rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/bench_gemm_shapes.py > /tmp/prof_gemm.log 2>&1
  t=$(find $out -name '*kernel_trace.csv' | head -1)
  echo "== R x K -> N = $sh" >> $out_txt
  python tools/gemm_trace_summary.py "$t" ${sh//x/ } >> $out_txt
done
