#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch).  usage: pmc_summary.py DIR > out.md"""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous" not in name:
            continue
        short = name.split("(anonymous namespace)::")[1].split("(")[0]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous" in name:
            short = name.split("(anonymous namespace)::")[1].split("(")[0]
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | dispatches | avg us (profiled) | counter | mean per dispatch |")
print("|---|---|---|---|---|")
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"| {k} | {len(v)} | {sum(dur[k])/max(len(dur[k]),1):.1f} | {c} | {sum(v)/len(v):.4g} |")
