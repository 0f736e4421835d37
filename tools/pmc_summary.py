#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per kernel name, average of each counter per dispatch."""
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/pass1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(k, [0])
    print(f"== {k}  n={len(d)} avg_us={sum(d)/max(1,len(d)):.1f}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} {sum(v)/len(v):16.1f}")
