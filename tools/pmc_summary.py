#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per (kernel name, grid size) -- the same template instantiation serves several
layers -- the average of each counter per dispatch and the average duration."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:70], r.get("Grid_Size", "?") + " lds=" + r.get("LDS_Block_Size", "?"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
seen = set()
for f in glob.glob(out + "/pass1/*/*counter_collection.csv"):     # durations: one entry per dispatch of the first pass
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in seen:
            continue
        seen.add(r["Dispatch_Id"])
        key = (r["Kernel_Name"][:70], r.get("Grid_Size", "?") + " lds=" + r.get("LDS_Block_Size", "?"))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
durn = {}
for k in sorted(acc, key=lambda k: -sum(dur.get(k, durn.get(k[0], [0])))):
    d = dur.get(k) or durn.get(k[0], [0])
    print(f"== {k[0]}  grid={k[1]}  n={len(d)} avg_us={sum(d) / max(1, len(d)):.1f}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} {sum(v) / len(v):16.1f}")
