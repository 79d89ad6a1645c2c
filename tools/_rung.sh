cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fused_ln_gpu.py tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_pipeline_gpu.py tests/test_attention.py -q -x > gpurun_out/ln_tests.txt 2>&1; tail -3 gpurun_out/ln_tests.txt
python bench.py --in-step-steps 0 > gpurun_out/bench_f1.json 2> gpurun_out/bench_f1.err
EDA_FUSED_LINEAR_LN=0 python bench.py --in-step-steps 0 > gpurun_out/bench_f0.json 2> gpurun_out/bench_f0.err
