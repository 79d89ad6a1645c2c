"""Per-queue busy time and idle gaps of a graph-replayed bench run from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 10 --warmup 3 --in-step-steps 0
    python tools/queue_gaps.py /tmp/kt 10

Takes the LAST `steps` replays: the window is cut at the last `steps` occurrences of the first kernel of the main queue's
period.  Prints, per queue: launches / step, busy ms / step, idle ms / step inside the window, the gap histogram and the
kernels with the most time."""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:64]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    for q in byq:
        byq[q].sort()
    # the timed region: the last `steps` x period launches of the busiest queue
    main_q = max(byq, key=lambda q: len(byq[q]))
    t_end = byq[main_q][-1][1]
    # period: find via autocorrelation of names at the tail
    names = [k[2] for k in byq[main_q]]
    per = None
    for p in range(200, 4000):
        if names[-p:] == names[-2 * p:-p]:
            per = p
            break
    if per is None:
        print("no period found"); return
    t0 = byq[main_q][-steps * per][0]
    print("main queue %s: period %d launches; window %.3f ms = %.3f ms / step" % (main_q, per, (t_end - t0) / 1e6, (t_end - t0) / 1e6 / steps))
    for q, L in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        W = [k for k in L if k[0] >= t0 and k[1] <= t_end]
        if not W:
            continue
        busy = sum(e - s for s, e, _ in W)
        gaps = [W[i + 1][0] - W[i][1] for i in range(len(W) - 1)]
        pos = [g for g in gaps if g > 0]
        print("queue %s: %.1f launches/step, busy %.3f ms/step, gaps %.3f ms/step (median %.2f us, >20us: %.3f ms/step in %d)" % (
            q, len(W) / steps, busy / 1e6 / steps, sum(pos) / 1e6 / steps, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0,
            sum(g for g in pos if g > 20000) / 1e6 / steps, sum(1 for g in pos if g > 20000)))
        hist = collections.Counter(min(int(g / 1000), 20) for g in pos)
        print("   gap histogram (us: count/step):", " ".join("%d:%.0f" % (k, v / steps) for k, v in sorted(hist.items())))
        agg = collections.defaultdict(lambda: [0, 0])
        for s, e, n in W:
            agg[n][0] += 1; agg[n][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
            print("   %8.3f ms/step %6.1f x  %7.2f us  %s" % (t / 1e6 / steps, c / steps, t / c / 1e3, n))


if __name__ == "__main__":
    main()
