#!/bin/bash
# One profiled eager run of the bench step + the default bench line -> gpurun_out/<tag>/ (then tools/summarize_profile.py).
# usage: tools/profile_run.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
tag=${1:-r03x}; shift
O=gpurun_out/$tag; mkdir -p $O
python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err
rm -rf /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o eager -- python bench.py --steps 20 --warmup 3 --graph 0 --cpu-scenes 0 "$@" > $O/bench_eager_under_rocprof.json 2> $O/bench_eager.err
find /tmp/pe -name "*kernel_stats.csv" -exec cp {} $O/bench_eager_kernel_stats_rocprofv3.csv \;
python tools/summarize_profile.py $tag 23 $O/bench_eager_kernel_stats_rocprofv3.csv $O/bench_default.json $O/bench_eager_under_rocprof.json > $O/summary.md 2> $O/summary.err
head -60 $O/summary.md
