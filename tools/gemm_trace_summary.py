"""Per-shape kernel durations from a rocprofv3 kernel trace of tools/bench_gemm_shapes.py (one process per shape)."""
import csv
import re
import sys
from collections import defaultdict

R, K, N = (int(v) for v in sys.argv[2:5])
rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name", "")
    if not re.search(r"gemm_|Cijk_|hipblaslt|rocblas", name):
        continue
    short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    short = re.sub(r"\(GemmArgs[^)]*\)$", "", short)[:72]
    acc[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flop = 2.0 * R * K * N
for k, v in acc.items():
    v = v[2:] if len(v) > 6 else v          # drop the warm-up launches
    s = sorted(v)
    avg = sum(v) / len(v)
    print(f"  {k:72s} n {len(v):3d} avg {avg:7.1f} us  med {s[len(s)//2]:7.1f}  min {s[0]:7.1f}   "
          f"{flop/avg/1e6:6.1f} TFLOP/s = {flop/avg/1e6/157.3:5.3f} of the fp32 MFMA peak")
