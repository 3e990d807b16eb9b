"""Sum rocprofv3 --pmc counter values per (kernel, counter) from the CSV output tree.

usage: pmc_sum.py <dir> [kernel-substring]
"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(float)
cnt = defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if flt not in k:
                continue
            k = k.split("(")[0][-60:]
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:32s} {v / cnt[(k, c)]:18.1f} per dispatch ({cnt[(k, c)]} dispatches)")
