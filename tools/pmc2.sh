#!/bin/bash
# Memory-path counters (TA / TD / vector L1 = TCP / L2 = TCC) for the per-kernel micro-benchmark, one rocprofv3 pass per
# counter group (PMC runs carry --kernel-trace only; at most two counters of a block per pass: more "exceeds the
# capabilities of the hardware" and the aborted profiler hangs).  usage: tools/pmc2.sh <outdir> [kbench args...]
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 $R/tools/kbench.py "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
