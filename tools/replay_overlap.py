"""Per-step count of back-to-back kernel pairs whose execution windows overlap, from a rocprofv3
kernel-trace CSV of a bench.py run (steps are delimited by the fused AdamW launches):
    rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --steps 8 --warmup 5 --kernel-steps 0 ...
    python tools/replay_overlap.py out/.../*_kernel_trace.csv"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    opt = [i for i, (_, _, k) in enumerate(ev) if "FusedOptimizer" in k]
    ends = []
    for i in opt:
        if not ends or i - ends[-1] > 5:
            ends.append(i)
        else:
            ends[-1] = i
    prev = 0
    for n, last in enumerate(ends):
        seg = ev[prev:last + 1]
        cnt = tot = worst = maxend = 0
        for s, e, _ in seg:
            if s < maxend:
                cnt += 1
                tot += maxend - s
                worst = max(worst, maxend - s)
            maxend = max(maxend, e)
        print("step %2d: %5d kernels, %8.2f ms, overlapping starts %4d, total %.1f us, worst %.2f us"
              % (n, len(seg), (seg[-1][1] - seg[0][0]) / 1e6, cnt, tot / 1e3, worst / 1e3))
        prev = last + 1


if __name__ == "__main__":
    main()
