cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_model_gpu.py tests/test_two_rank_gpu.py -m gpu -q -x > $O/pytest_attn.txt 2>&1; tail -6 $O/pytest_attn.txt
python tools/time_qproj_site.py 2>&1 | tail -6
for f in 1 0; do EDA_MHA_QPROJ=$f timeout 600 python bench.py --no-cpu-baseline --in-step-steps 0 --steps 20 > $O/bench_qproj$f.json 2> $O/bench_qproj$f.err; python -c "
import json;d=json.loads(open('$O/bench_qproj$f.json').read().strip().splitlines()[-1]);print('qproj=$f',d['value'],d['ms_per_step'])"; done
for f in 1 0; do EDA_MHA_QPROJ=$f timeout 600 python bench.py --no-cpu-baseline --in-step-steps 0 --steps 20 > $O/bench_qproj$f.json 2> $O/bench_qproj$f.err; python -c "
import json;d=json.loads(open('$O/bench_qproj$f.json').read().strip().splitlines()[-1]);print('qproj=$f',d['value'],d['ms_per_step'])"; done
