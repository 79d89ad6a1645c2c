cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python tools/measure_traffic.py > $O/traffic.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --fps-prefetch 0 > $O/bench_fps_in_step.json 2> $O/bench_fps_in_step.err
python bench.py --text-stream 0 > $O/bench_one_graph.json 2> $O/bench_one_graph.err
python bench.py --split-graphs > $O/bench_split_graphs.json 2> $O/bench_split_graphs.err
python bench.py --tokens 130 > $O/bench_130_tokens.json 2> $O/bench_130_tokens.err
python bench.py --loss hungarian > $O/bench_hungarian_loss.json 2> $O/bench_hungarian_loss.err
python bench.py --attn-dtype bf16 > $O/bench_attn_bf16.json 2> $O/bench_attn_bf16.err
python bench.py --attn-dtype f16 --tokens 130 > $O/bench_attn_f16_130_tokens.json 2> $O/bench_attn_f16_130_tokens.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o eager -- python bench.py --steps 20 --warmup 3 --graph 0 --cpu-scenes 0 > $O/bench_eager_under_rocprof.json 2> $O/bench_eager.err
find /tmp/pe -name "*kernel_stats.csv" -exec cp {} $O/bench_eager_kernel_stats_rocprofv3.csv \;
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02p/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
