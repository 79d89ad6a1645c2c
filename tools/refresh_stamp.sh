# After a change under eda_amd/csrc/: the GPU suite, the PMC traffic entries (stamped with the new source hash), the default bench
# line and the per-shape GEMM trace.  usage (GPU box): bash tools/refresh_stamp.sh <tag> -> gpurun_out/<tag>/, gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-refresh}; O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -1 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python tools/measure_traffic.py > $O/traffic.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tools/prof_gemm_shapes.sh ${tag} > /dev/null 2>&1
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["source_hash"])
PY
