#!/bin/bash
# rocprofv3 kernel durations of one row-GEMM shape under several tile configurations (EDA_GEMM_CFG), next to torch.mm
# usage: tools/bench_gemm_cfgs.sh R K N "cfg cfg ..."
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# cfg "d<id>" = gemm_dma_kernel configuration <id> of gemm.hip's launch_dma table (EDA_GEMM_DMA), "r<cfg>" =
# gemm_rows_kernel configuration (EDA_GEMM_CFG, DMA kernel off; r0 = its own selection), 0 = the library's default selection
R=$1; K=$2; N=$3; cfgs=${4:-"r0 d1 d2 d3 d4 d5 d6 d7 d8 d9 d10 d11 d12 d13 d14"}
for c in $cfgs; do
  out=/tmp/pg_$c; rm -rf $out
  unset EDA_GEMM_CFG EDA_GEMM_DMA EDA_GEMM_DMA_NST
  case $c in
    0) ;;
    d*) export EDA_GEMM_DMA=${c#d} ;;
    r0) export EDA_GEMM_DMA=0 ;;
    r*) export EDA_GEMM_CFG=${c#r}; export EDA_GEMM_DMA=0 ;;
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
    if "gemm_rows" in n or "Cijk" in n or "gemm_dma" in n:
        avg = float(r["AverageNs"]) / 1e3
        kind = "lib"
        if "gemm_rows" in n or "gemm_dma" in n:
            inst = n[n.index("<") + 1:n.rindex(">")].replace(" ", "").split(",")
            nn = inst[2] == "1" if "gemm_rows" in n else inst[4] == "1"
            kind = ("rows" if "gemm_rows" in n else "dma") + ("_dX" if nn else "_fwd")
        out.append("%s %.1f us (%.0f TF)" % (kind, avg, fl / avg / 1e6))
print("cfg", c, "|", " ; ".join(out))
PY
done
