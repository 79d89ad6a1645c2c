"""From a rocprofv3 kernel trace of the graph-replayed bench: per kernel (name, grid, workgroup) the number of ROUNDS its
workgroups need on 256 CUs given its LDS, register and wave footprint -- a launch with 1.1-1.6 rounds leaves most CUs idle for
its second round (round 5: the short-key attention backward, 384 workgroups where 256 fit at once: 54 -> 42 us).
    python tools/rounds_report.py <trace dir> <steps>"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:70]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
        lds = int(r["LDS_Block_Size"]); vg = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
        key = (short(r["Kernel_Name"]), grid, wg, lds, vg)
        agg[key][0] += 1
        agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = []
    for (name, grid, wg, lds, vg), (c, t) in agg.items():
        waves = (wg + 63) // 64
        per_simd = max(1, 512 // max(vg, 1)) if vg else 8
        per_simd = min(per_simd, 8)
        by_waves = (4 * per_simd) // waves if waves <= 4 * per_simd else 0
        by_lds = (160 * 1024) // lds if lds else 99
        per_cu = max(1, min(by_waves if by_waves else 1, by_lds, 32 // waves if waves <= 32 else 1))
        rounds = grid / (256.0 * per_cu)
        out.append((t / steps / 1e6, c / steps, t / c / 1e3, name, grid, wg, lds, vg, per_cu, rounds))
    out.sort(reverse=True)
    print("%8s %6s %8s  %-70s %7s %5s %7s %5s %6s %6s" % ("ms/step", "x", "us", "kernel", "WGs", "wg", "LDS", "regs", "WG/CU", "rounds"))
    for o in out[:70]:
        flag = "  <--" if (1.03 < o[9] < 1.7 or 2.03 < o[9] < 2.5) and o[2] > 8 else ""
        print("%8.3f %6.1f %8.2f  %-70s %7d %5d %7d %5d %6d %6.2f%s" % (o + (flag,))[:11] if False else
              "%8.3f %6.1f %8.2f  %-70s %7d %5d %7d %5d %6d %6.2f%s" % (o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], flag))


if __name__ == "__main__":
    main()
