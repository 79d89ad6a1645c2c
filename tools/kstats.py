"""Summarise a rocprofv3 kernel_stats.csv: name, calls, avg us, total ms (top N)."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
rows.sort(key=lambda r: -r[3])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for n, c, a, t in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{t / div:9.3f} ms  {c / div:7.1f} calls  {a:9.1f} us  {n[:110]}")
