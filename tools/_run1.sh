cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -30 $O/pytest_gpu.txt
timeout 600 python bench.py --gpus 2 --share-gpu --per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --in-step-steps 0 > $O/bench_2rank_share.json 2> $O/bench_2rank_share.err; tail -5 $O/bench_2rank_share.err; tail -c 600 $O/bench_2rank_share.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d.get('in_step'))"
