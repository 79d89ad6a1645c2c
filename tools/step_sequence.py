"""The main queue's launch SEQUENCE of one replayed training step, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 10 --warmup 3 --in-step-steps 0 --cpu-scenes 0
    python tools/step_sequence.py /tmp/kt [--native 0]

One line per launch of the LAST full period of the busiest queue: index, start offset (us), duration (us), idle gap in
front of it (us), N = a kernel of libeda_hip.so / T = anything else, and the kernel name.  With --native 0 only the T rows
(with the names of their neighbours).  The summary counts the T launches by name."""
import collections
import csv
import glob
import re
import sys

NATIVE = re.compile(r"^(adamw_flat|add_dropout_ln|add_n_kernel|ball_query|bf16x3_split|bn_|box_loss|center_query_pos|colsum|compact_targets|"
                    r"copy_kernel|det_scatter|dropout_flat|fps_|gather_points|gemm_|gq_|grad_sumsq|loss_combine|match_|pos_align|"
                    r"scale_by_scene|seed_objectness|sem_align|"
                    r"group_|l2norm_|linear_frozen|ln_reduce|lsa_|mha|peer_|rows_scatter|sa_|three_|tiny_out|transpose_batch|wcolsum|"
                    r"weight_transpose|wgrad|zero_kernel)")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return n[:150]


def main():
    d = sys.argv[1]
    only_t = "--native" in sys.argv and sys.argv[sys.argv.index("--native") + 1] == "0"
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    byq = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    q = max(byq, key=lambda k: len(byq[k]))
    rows = sorted(byq[q])
    names = [r[2] for r in rows]
    # period: distance between the last two occurrences of the rarest long kernel
    anchor = "wgrad_grouped_bf16x3_kernel"
    idx = [i for i, n in enumerate(names) if n.startswith(anchor)]
    if len(idx) < 3:
        print("anchor kernel not found"); return
    lo, hi = idx[-3] + 1, idx[-2] + 1          # one full period ending with the anchor
    seq = rows[lo:hi]
    t0 = seq[0][0]
    cnt = collections.Counter()
    tt = collections.Counter()
    for i, (s, e, n) in enumerate(seq):
        nat = bool(NATIVE.match(n))
        if not nat:
            cnt[n[:110]] += 1
            tt[n[:110]] += (e - s) / 1e3
        gap = (s - seq[i - 1][1]) / 1e3 if i else 0.0
        if only_t and nat:
            continue
        ctx = ""
        if only_t:
            ctx = "   after: " + re.sub(r"<.*", "", seq[i - 1][2])[:28] + " | before: " + (re.sub(r"<.*", "", seq[i + 1][2])[:28] if i + 1 < len(seq) else "")
        print("%4d %9.1f %7.1f %6.1f %s %s%s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, "N" if nat else "T", n[:110], ctx))
    print("# period: %d launches, %.3f ms; not libeda_hip.so: %d launches, %.3f ms" % (len(seq), (seq[-1][1] - t0) / 1e6, sum(cnt.values()), sum(tt.values()) / 1e3))
    for n, c in cnt.most_common():
        print("# %3d x %7.1f us total  %s" % (c, tt[n], n))


if __name__ == "__main__":
    main()
