# The evidence run of a round: tests, smoke, PMC traffic, the bench lines of BASELINE.json's single-GPU configurations and
# one profiled eager run.  usage (on the GPU box): bash tools/final_run.sh <tag>   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r04z}
O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python tools/measure_traffic.py > $O/traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # (this box's copy: the bench lines below carry the traffic of THIS build)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --fps-prefetch 0 --in-step-steps 0 > $O/bench_fps_in_step.json 2> $O/bench_fps_in_step.err
python bench.py --fps-prefetch 1 --in-step-steps 0 > $O/bench_sa1_prefetch_only.json 2> $O/bench_sa1_prefetch_only.err
python bench.py --text-stream 0 > $O/bench_one_graph.json 2> $O/bench_one_graph.err
python bench.py --split-graphs --in-step-steps 0 > $O/bench_split_graphs.json 2> $O/bench_split_graphs.err
python bench.py --tokens 130 --in-step-steps 0 > $O/bench_130_tokens.json 2> $O/bench_130_tokens.err
python bench.py --loss hungarian --in-step-steps 0 > $O/bench_hungarian_loss.json 2> $O/bench_hungarian_loss.err
python bench.py --attn-dtype bf16 --in-step-steps 0 > $O/bench_attn_bf16.json 2> $O/bench_attn_bf16.err
python bench.py --attn-dtype f16 --tokens 130 --in-step-steps 0 > $O/bench_attn_f16_130_tokens.json 2> $O/bench_attn_f16_130_tokens.err
EDA_FAST_ROBERTA=0 python bench.py --in-step-steps 0 > $O/bench_stock_roberta.json 2> $O/bench_stock_roberta.err
python bench.py --batches 1 --in-step-steps 0 > $O/bench_one_batch.json 2> $O/bench_one_batch.err
# round 4: the sampler policies, the N > 1 step structure through a one-rank RCCL group, the fused q-projection sites
EDA_FPS_BUCKET=1 python bench.py --in-step-steps 0 > $O/bench_fps_bucket.json 2> $O/bench_fps_bucket.err
EDA_FPS_BUCKET=1 python bench.py --fps-prefetch 0 --in-step-steps 0 > $O/bench_fps_bucket_in_step.json 2> $O/bench_fps_bucket_in_step.err
python bench.py --force-dist --in-step-steps 0 > $O/bench_force_dist.json 2> $O/bench_force_dist.err
python bench.py --force-dist --overlap-allreduce 1 --in-step-steps 0 > $O/bench_force_dist_overlap.json 2> $O/bench_force_dist_overlap.err
python bench.py --force-dist --sync-bn native --in-step-steps 0 > $O/bench_force_dist_sync_bn_native.json 2> $O/bench_force_dist_sync_bn_native.err
python bench.py --force-dist --sync-bn collective --in-step-steps 0 > $O/bench_force_dist_sync_bn_collective.json 2> $O/bench_force_dist_sync_bn_collective.err
# round 5: ordered scatter sums, split contraction / key split / bf16x3 forward off, the races' reproducer, queue gaps
python bench.py --deterministic 1 --in-step-steps 0 > $O/bench_deterministic.json 2> $O/bench_deterministic.err
EDA_GEMM_SPLITK=0 python bench.py --in-step-steps 0 > $O/bench_splitk_off.json 2> $O/bench_splitk_off.err
EDA_GEMM_KC96=0 python bench.py --in-step-steps 0 > $O/bench_kc96_off.json 2> $O/bench_kc96_off.err
EDA_TIMING_SKIP_SIDE=fps,text python bench.py --in-step-steps 0 > $O/bench_main_stream_alone_TIMING_ONLY.json 2> /dev/null
EDA_MHA3=0 EDA_MHA2_KSPLIT=0 python bench.py --in-step-steps 0 > $O/bench_mha3_ksplit_off.json 2> $O/bench_mha3_ksplit_off.err
python tools/dbg_pipeline_gemm.py 12 > $O/dbg_pipeline_gemm.txt 2>&1
python tools/bench_sa_eval.py > $O/sa_eval.txt 2>&1
EDA_MHA_QPROJ=auto python bench.py --in-step-steps 0 > $O/bench_qproj_auto.json 2> $O/bench_qproj_auto.err
EDA_WGRAD_BF16X3=0 python bench.py --in-step-steps 0 > $O/bench_wgrad_fp32_mfma.json 2> $O/bench_wgrad_fp32_mfma.err
EDA_BATCHED_HEADS=0 python bench.py --in-step-steps 0 > $O/bench_heads_per_head.json 2> $O/bench_heads_per_head.err
EDA_RESIDUAL_LINK=0 python bench.py --in-step-steps 0 > $O/bench_residual_link_off.json 2> $O/bench_residual_link_off.err
EDA_GEMM_B3ROWS=0 python bench.py --in-step-steps 0 > $O/bench_b3rows_off.json 2> $O/bench_b3rows_off.err
# round 6: in-launch merge of the split attention backward off, the frozen text encoder on fp32-MFMA products, the key-per-wave forward,
# the training loss in its element-wise torch form
EDA_MHA2_BWD_MERGE=0 EDA_MHA2_BWD_DBUF=0 python bench.py --in-step-steps 0 > $O/bench_mha_bwd_r05_form.json 2> $O/bench_mha_bwd_r05_form.err
EDA_FROZEN_B3=0 python bench.py --in-step-steps 0 > $O/bench_frozen_b3_off.json 2> $O/bench_frozen_b3_off.err
EDA_MHA4=1 python bench.py --in-step-steps 0 > $O/bench_mha4_on.json 2> $O/bench_mha4_on.err
python bench.py --force-dist --loss hungarian --in-step-steps 0 > $O/bench_force_dist_hungarian.json 2> $O/bench_force_dist_hungarian.err
python bench.py --force-dist --sync-bn native --loss hungarian --in-step-steps 0 > $O/bench_force_dist_sync_bn_native_hungarian.json 2> $O/bench_force_dist_sync_bn_native_hungarian.err
python bench.py --text-encoder-mode eval --in-step-steps 0 > $O/bench_text_encoder_eval.json 2> $O/bench_text_encoder_eval.err
EDA_FUSED_LOSS=0 python bench.py --loss hungarian --in-step-steps 0 > $O/bench_hungarian_loss_torch_form.json 2> $O/bench_hungarian_loss_torch_form.err
python tools/loss_census.py > $O/loss_census.txt 2>&1
EDA_FUSED_LOSS=0 python tools/loss_census.py > $O/loss_census_torch_form.txt 2>&1
python tools/loss_phase_profile.py > $O/loss_phase_profile.txt 2>&1
python tools/bench_gemm_frozen.py > $O/gemm_frozen.txt 2>&1
python tools/bench_gemm_b3rows.py > $O/gemm_b3rows_tool.txt 2>&1
python tools/bench_wgrad_grouped.py > $O/wgrad_grouped.txt 2>&1
python tools/time_qproj_site.py > $O/qproj_site.txt 2>&1
python tools/time_linear_ln.py 2048 288 > $O/linear_ln.txt 2>&1
python tools/fps_handoffs.py 8 50000 2048 > $O/fps_cluster.txt 2>&1
EDA_FPS_BUCKET=1 python tools/fps_handoffs.py 8 50000 2048 > $O/fps_bucket.txt 2>&1
rm -rf /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o eager -- python bench.py --steps 20 --warmup 3 --graph 0 --cpu-scenes 0 > $O/bench_eager_under_rocprof.json 2> $O/bench_eager.err
find /tmp/pe -name "*kernel_stats.csv" -exec cp {} $O/bench_eager_kernel_stats_rocprofv3.csv \;
python tools/summarize_profile.py $tag 23 $O/bench_eager_kernel_stats_rocprofv3.csv $O/bench_default.json $O/bench_eager_under_rocprof.json > $O/summary.md 2> $O/summary.err
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 10 --warmup 3 --in-step-steps 0 --cpu-scenes 0 > $O/bench_under_trace.json 2>/dev/null
python tools/queue_gaps.py /tmp/kt 10 > $O/queue_gaps.txt 2>&1
python tools/step_sequence.py /tmp/kt > $O/step_sequence.txt 2>&1
python tools/step_sequence.py /tmp/kt --native 0 > $O/step_sequence_not_native.txt 2>&1
tools/prof_gemm_shapes.sh ${tag} > /dev/null 2>&1
tools/prof_mha.sh ${tag}_f32 > /dev/null 2>&1
ATTN_DTYPE=bf16 tools/prof_mha.sh ${tag}_bf16 > /dev/null 2>&1
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
