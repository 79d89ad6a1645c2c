"""From a rocprofv3 --kernel-trace CSV of `bench.py` (graph replays): duration of every furthest-point-sampling
launch of SA1 and how much of the frozen text encoder's kernels ran inside its window.
    python tools/overlap_trace.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", ""))) for r in rows]
ks.sort()
fps = [k for k in ks if "fps_spec_kernel" in k[2]]
print("fps launches", len(fps), "queues", sorted({k[3] for k in ks}))
for s, e, n, q in fps[-6:]:
    inside = [k for k in ks if k[3] != q and k[0] < e and k[1] > s]
    busy = sum(min(k[1], e) - max(k[0], s) for k in inside)
    print(f"fps {(e - s) / 1e3:8.1f} us on queue {q}; other-queue kernels inside: {len(inside)} busy {busy / 1e3:8.1f} us")
# step period
if len(fps) > 3:
    print("period between fps starts (us):", [round((fps[i + 1][0] - fps[i][0]) / 1e3, 1) for i in range(len(fps) - 4, len(fps) - 1)])
# timeline of the non-fps queues around the second-to-last fps launch
if len(fps) > 2:
    s, e, n, q = fps[-3]
    oth = [k for k in ks if k[3] != q and s - 30_000_000 < k[0] < e + 30_000_000]
    print("other-queue kernels within +-30 ms of that fps start:", len(oth))
    for k in oth[:: max(1, len(oth) // 40)]:
        print(f"  q{k[3]} start {(k[0] - s) / 1e3:10.1f} us  dur {(k[1] - k[0]) / 1e3:7.1f}  {k[2][:60]}")
    main = [k for k in ks if k[3] == q and s - 2_300_000 < k[0] < s + 200_000]
    print("main-queue kernels in the 2.3 ms before that fps start:", len(main), "busy us", sum(k[1] - k[0] for k in main[:-1]) / 1e3)
    for k in main[:: max(1, len(main) // 40)] + main[-6:]:
        print(f"  MAIN q{k[3]} start {(k[0] - s) / 1e3:10.1f} us  dur {(k[1] - k[0]) / 1e3:7.1f}  {k[2][:60]}")
