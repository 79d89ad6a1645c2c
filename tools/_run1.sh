cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_two_rank_gpu.py tests/test_sync_bn_gpu.py tests/test_graph_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x > $O/pytest_gpu2.txt 2>&1; tail -5 $O/pytest_gpu2.txt
for v in "" "--sync-bn" "--overlap-allreduce 0"; do
timeout 600 python bench.py --force-dist $v --no-cpu-baseline --steps 10 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; grep "bench +" $O/bench_force_dist.err | tail -3; python -c "
import json;d=json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1]);print('$v',d['value'],d['ms_per_step'],d['config']['parallelism'],'|',d['config']['gradient_allreduce'],'|',d['config']['batchnorm'],d['loss'])"
done
OPS='*' MIN=1 timeout 600 python tools/op_census.py > $O/census.txt 2>&1; head -3 $O/census.txt
