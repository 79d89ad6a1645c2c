cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests/test_grouped_gpu.py tests/test_sa_cl_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_edge_cases_gpu.py tests/test_wgrad_queue_gpu.py -m gpu -q -x > $O/pytest_bn.txt 2>&1; tail -8 $O/pytest_bn.txt
for f in 1 0 1 0; do EDA_BN_ROWSPLIT=$f timeout 600 python bench.py --no-cpu-baseline --in-step-steps 0 --steps 20 > $O/bench_bn$f.json 2> $O/bench_bn$f.err; python -c "
import json;d=json.loads(open('$O/bench_bn$f.json').read().strip().splitlines()[-1]);print('rowsplit=$f',d['value'],d['ms_per_step'], [ (k['op'],k['dims'][:3],round(k['ms']*1e3,1)) for k in d['kernels'] if k['op'].startswith('bn_relu') and k['dims'][0]==2048][:6])"; done
for r in 8 32; do EDA_BN_ROWSPLIT_RPB=$r timeout 600 python bench.py --no-cpu-baseline --in-step-steps 0 --steps 20 > $O/bench_bnr.json 2> $O/bench_bnr.err; python -c "
import json;d=json.loads(open('$O/bench_bnr.json').read().strip().splitlines()[-1]);print('rpb=$r',d['value'],d['ms_per_step'], [ (k['op'],k['dims'][:3],round(k['ms']*1e3,1)) for k in d['kernels'] if k['op'].startswith('bn_relu') and k['dims'][0]==2048][:6])"; done
