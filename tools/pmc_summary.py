#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel from a counter_collection CSV: python tools/pmc_summary.py file.csv"""
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")[:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
print(f"{'kernel':60s} " + " ".join(f"{n[:18]:>18s}" for n in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    if not k.startswith("cmbl"): continue
    print(f"{k:60s} " + " ".join(f"{acc[k][n]/max(cnt[k][n],1):18.1f}" for n in names))
