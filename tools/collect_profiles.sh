#!/bin/bash
# Copy the outputs of tools/final_run.sh <tag> (gpurun_out/<tag>/) into profiles/ under the round's names.
# usage: tools/collect_profiles.sh <tag> <round prefix, e.g. r03>
cd "$(dirname "$0")/.." || exit 1
tag=$1; r=$2; O=gpurun_out/$tag
python tools/summarize_profile.py $r 23 $O/bench_eager_kernel_stats_rocprofv3.csv $O/bench_default.json $O/bench_eager_under_rocprof.json > profiles/${r}_summary.md
for f in default fps_in_step sa1_prefetch_only one_graph split_graphs 130_tokens hungarian_loss hungarian_loss_torch_form force_dist_hungarian force_dist_sync_bn_native_hungarian text_encoder_eval attn_bf16 attn_f16_130_tokens stock_roberta one_batch eager_under_rocprof fps_bucket fps_bucket_in_step force_dist force_dist_overlap force_dist_sync_bn_native force_dist_sync_bn_collective deterministic splitk_off kc96_off main_stream_alone_TIMING_ONLY mha3_ksplit_off qproj_off qproj_auto wgrad_fp32_mfma heads_per_head residual_link_off b3rows_off mha_bwd_r05_form frozen_b3_off mha4_on; do
  [ -f $O/bench_$f.json ] && cp $O/bench_$f.json profiles/${r}_bench_$f.json
done
for f in qproj_site linear_ln fps_cluster fps_bucket wgrad_grouped queue_gaps dbg_pipeline_gemm sa_eval step_sequence_not_native gemm_frozen loss_census loss_census_torch_form loss_phase_profile; do [ -f $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > profiles/${r}_$f.txt; done
cp $O/bench_eager_kernel_stats_rocprofv3.csv profiles/${r}_bench_eager_kernel_stats_rocprofv3.csv
cp gpurun_out/mha_${tag}_f32.txt profiles/${r}_mha_f32.txt
cp gpurun_out/mha_${tag}_bf16.txt profiles/${r}_mha_bf16.txt
[ -f gpurun_out/gemm_shapes_${tag}.txt ] && cp gpurun_out/gemm_shapes_${tag}.txt profiles/${r}_gemm_shapes.txt
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
tail -3 $O/pytest_gpu.txt > profiles/${r}_pytest_gpu_tail.txt
