cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_sa_fused_gpu.py -m gpu -q -x -k "full_size_gradients" > $O/pytest_fs.txt 2>&1; tail -15 $O/pytest_fs.txt
timeout 900 python -m pytest tests/test_two_rank_gpu.py -m gpu -q -x > $O/pytest_2r.txt 2>&1; tail -4 $O/pytest_2r.txt
