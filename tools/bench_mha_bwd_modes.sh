#!/bin/bash
# attention backward: in-launch merge of the split ranges (EDA_MHA2_BWD_MERGE=1) against the second launch (=0), and the
# short-key variants with double-buffered query chunks (EDA_MHA2_BWD_DBUF=1) -- rocprofv3 kernel durations per shape.
# usage: tools/bench_mha_bwd_modes.sh -> gpurun_out/mha_bwd_modes.txt
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/mha_bwd_modes.txt
: > $out
for cfg in "EDA_MHA2_BWD_MERGE=0" "EDA_MHA2_BWD_MERGE=1" "EDA_MHA2_BWD_MERGE=0 EDA_MHA2_BWD_DBUF=1"; do
  echo "#### $cfg" >> $out
  env $cfg bash tools/prof_mha.sh modes_tmp > /dev/null 2>&1
  grep -E "^==|bwd_kernel|part_reduce|backward total" gpurun_out/mha_modes_tmp.txt >> $out
done
rm -f gpurun_out/mha_modes_tmp.txt
