"""Average durations of the attention kernels in a rocprofv3 kernel-trace CSV (one shape per trace)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name", "")
    if "mha" not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    short = re.sub(r"\((Mha2Args|MhaArgs|Mha16Args)[^)]*\)$", "", short)[:60]
    acc[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in acc.items():
    v = v[1:] if len(v) > 3 else v          # drop the warm-up launch
    s = sorted(v)
    fwd = "fwd" in k
    print(f"  {k:60s} n {len(v):3d} avg {sum(v)/len(v):8.1f} us  med {s[len(s)//2]:8.1f}  min {s[0]:8.1f}")
    if not fwd:
        tot += sum(v) / len(v)
print(f"  backward total (sum of avgs) {tot:8.1f} us")
