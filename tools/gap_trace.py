"""Gaps between consecutive kernels of the busiest queue in a rocprofv3 --kernel-trace CSV of `bench.py` (graph replays):
how much of a step is idle time between launches, and where the largest gaps are.
    python tools/gap_trace.py kernel_trace.csv [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ks.sort()
byq = collections.defaultdict(list)
for k in ks:
    byq[k[3]].append(k)
print({q: len(v) for q, v in byq.items()})
q = max(byq, key=lambda x: len(byq[x]))
main = byq[q]
# the timed region: the last `steps` occurrences of the optimizer kernel mark step ends
ends = [i for i, k in enumerate(main) if "multi_tensor_apply" in k[2]]
per = len(ends) // max(1, (len(ends) // max(steps, 1)))
marks = ends[-1::-max(1, len(ends) // (steps + 3))][:steps + 1][::-1] if ends else []
if len(marks) >= 3:
    a, b = marks[1], marks[-1]
    seg = main[a:b]
    nst = len(marks) - 2
    span = (seg[-1][1] - seg[0][0]) / 1e3 / nst
    busy = sum(k[1] - k[0] for k in seg) / 1e3 / nst
    gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(seg) / nst:.0f} kernels/step, span {span:.1f} us/step, busy {busy:.1f}, gaps {sum(g[0] for g in pos) / 1e3 / nst:.1f} "
          f"(overlapping pairs: {sum(1 for g in gaps if g[0] <= 0) / nst:.0f})")
    hist = collections.Counter()
    for g in pos:
        hist[min(int(g[0] / 1000), 20)] += 1
    print("gap histogram (us -> count/step):", {k: round(v / nst, 1) for k, v in sorted(hist.items())})
    agg = collections.defaultdict(lambda: [0, 0.0])
    for g in pos:
        if g[0] > 3000:
            agg[(g[1], g[2])][0] += 1
            agg[(g[1], g[2])][1] += g[0] / 1e3
    for (p, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / nst:7.1f} us/step x{c / nst:5.1f}  {p}  ->  {n}")
