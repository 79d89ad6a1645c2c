cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_grouped_gpu.py tests/test_roberta_fast_gpu.py tests/test_fused_ln_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py -q -x > gpurun_out/gemm_tests.txt 2>&1; tail -3 gpurun_out/gemm_tests.txt
