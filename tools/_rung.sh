cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --in-step-steps 0 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python bench.py --in-step-steps 0 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_grouped_gpu.py tests/test_sa_cl_gpu.py tests/test_sa_fused_gpu.py -q -x > gpurun_out/gemm_tests.txt 2>&1; tail -3 gpurun_out/gemm_tests.txt
