"""For each GPU kernel in a trace from tools/trace_step.py: the chain of cpu ops that enclosed
its launch.  Aggregates (kernel short name, op chain) -> count, device time.
    python tools/analyze_trace.py trace.json.gz [kernel-substring ...]"""
import bisect
import collections
import gzip
import json
import re
import sys


def main():
    ev = json.load(gzip.open(sys.argv[1]))["traceEvents"]
    pats = sys.argv[2:]
    cpu = [e for e in ev if e.get("cat") in ("cpu_op", "user_annotation") and e.get("ph") == "X"]
    rt = {e["args"]["correlation"]: e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver")
          and "correlation" in e.get("args", {})}
    kern = [e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
    by_tid = collections.defaultdict(list)
    for e in cpu:
        by_tid[e["tid"]].append(e)
    for l in by_tid.values():
        l.sort(key=lambda e: e["ts"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k in kern:
        name = re.sub(r"at::native::|void |\(anonymous namespace\)::", "", k["name"])[:60]
        if pats and not any(p in k["name"] for p in pats):
            continue
        r = rt.get(k["args"].get("correlation"))
        chain = []
        if r is not None:
            l = by_tid.get(r["tid"], [])
            t = r["ts"]
            for e in l:
                if e["ts"] > t:
                    break
                if e["ts"] <= t <= e["ts"] + e["dur"]:
                    chain.append(e["name"])
        key = (name, " > ".join(c[:40] for c in chain[-4:]))
        agg[key][0] += 1
        agg[key][1] += k["dur"]
    tot = sum(v[1] for v in agg.values())
    print(f"total {tot / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} kernels")
    for (name, chain), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
        print(f"{t / 1e3:7.3f} ms x{c:4d}  {name:60s} | {chain}")


if __name__ == "__main__":
    main()
