cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "fps" > $O/pytest_fps.txt 2>&1; tail -5 $O/pytest_fps.txt
EDA_FPS_BUCKET=1 timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -1
EDA_FPS_BUCKET=1 EDA_FPS_BUCKET_NW=16 timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -1
EDA_FPS_BUCKET=1 EDA_FPS_BUCKET_NW=12 timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -1
timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -1
