"""Per-kernel averages of rocprofv3 --pmc counters from the CSV output tree, plus MFMA utilisation when the pass
holds SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE *
1024 SIMDs), the expression rocprofv3 -L lists for the derived counter).

usage: pmc_sum.py <dir> [kernel-substring]
"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
SIMDS = 256 * 4
XCDS = 8          # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (149.9 M "cycles" for a 9.6 ms kernel = 8 x 1.95 GHz):
                  # chip cycles = GRBM_GUI_ACTIVE / 8


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    return re.sub(r"\(.*", "", name)[:64]


acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", ""))
            if flt not in k:
                continue
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
kernels = sorted({k for k, _ in acc}, key=lambda k: -acc.get((k, "GRBM_GUI_ACTIVE"), acc.get((k, "SQ_BUSY_CYCLES"), 0.0)))
counters = sorted({c for _, c in acc})
print("kernel,dispatches," + ",".join(f"avg_{c}" for c in counters) + ",mfma_util_pct,valu_active_pct")
for k in kernels[:40]:
    n = max(cnt[(k, c)] for c in counters if (k, c) in cnt)
    avg = {c: acc[(k, c)] / cnt[(k, c)] for c in counters if (k, c) in cnt}
    util = valu = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
        util = f"{100.0 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / XCDS * SIMDS):.1f}"
    if "SQ_ACTIVE_INST_VALU" in avg and avg.get("GRBM_GUI_ACTIVE"):      # quad-cycles of VALU issue per SIMD-cycle
        valu = f"{100.0 * 4.0 * avg['SQ_ACTIVE_INST_VALU'] / (avg['GRBM_GUI_ACTIVE'] / XCDS * SIMDS):.1f}"
    print(f"\"{k}\",{n}," + ",".join(f"{avg.get(c, 0.0):.0f}" for c in counters) + f",{util},{valu}")
