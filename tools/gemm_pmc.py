"""PMC counters of one row-GEMM shape (tools/bench_gemm_one.py) per kernel: one rocprofv3 pass per counter.
usage: EDA_GEMM_DMA=<cfg> python tools/gemm_pmc.py R K N [counter ...]
FETCH_SIZE is printed in MB with the gfx950 correction (x2, KiB units) of MI355X_MICROARCH.md."""
import csv, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, K, N = sys.argv[1:4]
counters = sys.argv[4:] or ["FETCH_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ACTIVE", "SQ_BUSY_CYCLES"]
for c in counters:
    d = tempfile.mkdtemp(prefix="pmc_")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "tools/bench_gemm_one.py"), R, K, N], capture_output=True, cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"))
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if r.returncode != 0 or not fs:
        print(c, "unavailable"); continue
    tot, n = {}, {}
    for row in csv.DictReader(open(fs[0])):
        if row["Counter_Name"] != c: continue
        k = row["Kernel_Name"]
        if "gemm" not in k and "Cijk" not in k: continue
        k = k[:70]
        tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    for k in tot:
        v = tot[k] / n[k]
        if c == "FETCH_SIZE": print(f"{c:24s} {v * 2 * 1024 / 1e6:10.1f} MB   {k}")
        else: print(f"{c:24s} {v:14.0f}   {k}")
