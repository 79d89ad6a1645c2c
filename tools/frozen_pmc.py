"""PMC counters of csrc/gemm_frozen.hip's kernel on the text encoder's 768 -> 3072 shape (640 rows) next to the fp32-MFMA kernel:
one rocprofv3 --pmc pass per counter group (with --kernel-trace only).  usage: python tools/frozen_pmc.py"""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = '''
import sys, torch
sys.path.insert(0, %r)
from eda_amd import gemm
x = torch.randn(640, 768, device="cuda"); w = torch.randn(3072, 768, device="cuda") * 0.05; b = torch.randn(3072, device="cuda")
pl = gemm.frozen_planes(w)
for _ in range(12):
    gemm.linear_frozen(x, pl, b, act=2)
    gemm.linear_fwd(x, w, b, relu=2)
torch.cuda.synchronize()
''' % ROOT
GROUPS = [["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"]]
with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
    f.write(DRIVER)
    script = f.name
for grp in GROUPS:
    d = tempfile.mkdtemp(prefix="pmc_")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--", sys.executable, script],
                       capture_output=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if r.returncode != 0 or not fs:
        print(grp, "unavailable", r.stderr.decode()[-300:])
        continue
    tot, n = {}, {}
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "linear_frozen" not in k and "gemm_dma" not in k:
            continue
        key = (k[:60], row["Counter_Name"])
        tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
        n[key] = n.get(key, 0) + 1
    for (k, c) in sorted(tot):
        v = tot[(k, c)] / n[(k, c)]
        if c == "FETCH_SIZE":
            print("%-62s %-26s %12.2f MB per launch (x2 x KiB, gfx950 correction)" % (k, c, v * 2 * 1024 / 1e6))
        elif c == "WRITE_SIZE":
            print("%-62s %-26s %12.2f MB per launch" % (k, c, v * 1024 / 1e6))
        else:
            print("%-62s %-26s %14.0f per launch" % (k, c, v))
