#!/usr/bin/env python3
"""Idle time between kernels inside the timed steps of a bench run, from a rocprofv3 --kernel-trace CSV.
usage: tools/gaps.py <kernel_trace.csv> [steps]   (the last `steps` pack_btchw launches delimit the steps)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if e[2].startswith("void pack_btchw_rows_kernel")]
starts = starts[-steps - 1:] if len(starts) > steps else starts
tot_busy = tot_gap = 0
ngap = 0
big = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    busy = sum(e[1] - e[0] for e in seg)
    span = ev[b][0] - seg[0][0]
    gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)] + [(ev[b][0] - seg[-1][1], seg[-1][2][:50], "next step")]
    tot_busy += busy; tot_gap += span - busy; ngap += len(gaps)
    big += sorted(gaps, reverse=True)[:3]
n = len(starts) - 1
# per-kernel time per step (template arguments kept: they tell the layers apart).  The FIRST launch of a kernel in a step is
# kept apart: for the gate kernels it is time step 0, which has no h half of K (half the work of the other launches)
per = {}
for a, b in zip(starts[:-1], starts[1:]):
    seen = set()
    for e in ev[a:b]:
        nm = e[2].split("(")[0][:60]
        t = per.setdefault(nm, [0, 0, 0, 0])
        t[0] += 1; t[1] += e[1] - e[0]
        if nm not in seen:
            seen.add(nm)
            t[2] += 1; t[3] += e[1] - e[0]
for nm, (cnt, ns, c1, ns1) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    if ns / n / 1e3 >= 5:
        rest = f"   first launch of the step {ns1 / c1 / 1e3:7.1f} us, the others {(ns - ns1) / (cnt - c1) / 1e3:7.1f} us" if cnt > c1 else ""
        print(f"  {nm:60s} {cnt / n:6.1f} launches  {ns / cnt / 1e3:8.1f} us each  {ns / n / 1e3:8.1f} us per step{rest}")
print(f"{n} steps: {tot_busy / n / 1e3:.1f} us busy + {tot_gap / n / 1e3:.1f} us idle per step, {ngap // n} launches per step, mean gap {tot_gap / max(ngap, 1) / 1e3:.2f} us")
for g in sorted(big, reverse=True)[:8]:
    print(f"  gap {g[0] / 1e3:7.1f} us after {g[1]} -> {g[2]}")
