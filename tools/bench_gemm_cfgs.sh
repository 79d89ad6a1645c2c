#!/bin/bash
# rocprofv3 kernel durations of one row-GEMM shape under several tile configurations (EDA_GEMM_CFG), next to torch.mm
# usage: tools/bench_gemm_cfgs.sh R K N "cfg cfg ..."
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# cfg "g<MW><NW>" = gemm2.hip with that wave tile (EDA_GEMM2), a number = gemm_rows_kernel configuration (EDA_GEMM_CFG,
# gemm2 off), 0 = the library's default selection
R=$1; K=$2; N=$3; cfgs=${4:-"0 g12 g13 g14 g22 g23 g24 g43 g44"}
for c in $cfgs; do
  out=/tmp/pg_$c; rm -rf $out
  unset EDA_GEMM_CFG EDA_GEMM2
  case $c in
    0) ;;
    g*) export EDA_GEMM2=${c#g} ;;
    *) export EDA_GEMM_CFG=$c; export EDA_GEMM2=0 ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python tools/bench_gemm_one.py $R $K $N > /dev/null 2>&1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python - "$f" "$c" "$R" "$K" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
c, R, K, N = sys.argv[2], *[int(v) for v in sys.argv[3:6]]
fl = 2.0 * R * K * N
out = []
for r in rows:
    n = r["Name"]
    if "gemm_rows" in n or "Cijk" in n or "gemm2" in n:
        avg = float(r["AverageNs"]) / 1e3
        out.append("%s %.1f us (%.0f TF)" % ("rows" if "gemm_rows" in n else "gemm2" if "gemm2" in n else "lib", avg, fl / avg / 1e6))
print("cfg", c, "|", " ; ".join(out))
PY
done
