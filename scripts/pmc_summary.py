#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel sums."""
import csv, glob, os, sys, collections, re

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)
ncall = collections.defaultdict(int)

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]

for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for f in glob.glob(os.path.join(root, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        ncall[k] += 1
names = sorted(dur, key=lambda k: -dur[k])
ctrs = sorted({c for k in agg for c in agg[k]})
print("kernel,calls,total_us,avg_us," + ",".join(ctrs))
for k in names:
    row = [k, str(ncall[k]), "%.1f" % dur[k], "%.1f" % (dur[k] / max(ncall[k], 1))]
    for c in ctrs:
        n = calls[k].get(c, 0)
        row.append("%.4g" % (agg[k][c] / n) if n else "")
    print(",".join(row))
