#!/bin/bash
# cache-side counters for the per-kernel micro-benchmark. usage: tools/pmc2.sh <outdir> [kbench args...]
set -e
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 $R/tools/kbench.py "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
