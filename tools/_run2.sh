cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
EDA_FPS_BUCKET=1 EDA_HIP_LIB=$PWD/eda_amd/csrc/libeda_hip_fpsprof.so timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -3
EDA_FPS_BUCKET=1 EDA_FPS_BUCKET_NW=8 EDA_HIP_LIB=$PWD/eda_amd/csrc/libeda_hip_fpsprof.so timeout 300 python tools/fps_handoffs.py 8 50000 2048 2>&1 | tail -3
