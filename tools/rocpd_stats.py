#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the *_results.db written by --kernel-trace).
    python tools/rocpd_stats.py gpurun_out/prof_x/x_results.db [steps]
Prints microseconds per step (total / steps), launch count and average duration per kernel."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d "
         f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    tot = 0.0
    print(f"{'us/step':>10} {'calls':>6} {'avg_us':>9}  kernel")
    for name, n, su, av in cur.execute(q):
        tot += su
        print(f"{su / steps:10.1f} {n:6d} {av:9.1f}  {name[:120]}")
    print(f"{tot / steps:10.1f} total")


if __name__ == "__main__":
    main()
