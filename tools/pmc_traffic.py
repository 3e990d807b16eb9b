#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: calls and summed counter.
usage: pmc_traffic.py <dir-with-csv> <COUNTER>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    k = re.sub(r"\(.*", "", k)[:70]
    agg[k] += float(r["Counter_Value"]); cnt[k] += 1
print(f"counter,{sys.argv[2]}")
print("kernel,dispatches,sum,avg_per_dispatch")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:400]:      # (not truncated to 25 any more: make_traffic_json sums families from this table)
    print(f"\"{k}\",{cnt[k]},{v:.0f},{v / cnt[k]:.1f}")
