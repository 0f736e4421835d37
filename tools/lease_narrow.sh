#!/bin/bash
# Phase timelines of the narrow-layer launches (stamp build) + a B=1 step breakdown.  usage: tools/lease_narrow.sh <tag>
TAG=${1:-narrow}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for k in fwd1 fwd2 dgrad1 fused2 fwd0; do
  timeout -k 10 120 python tools/clockprobe.py --kernel $k --seconds 1 > "$OUT/clock_$k.txt" 2>&1 || exit 1
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 > "$OUT/stats.log" 2>&1 )
TR=$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/gaps.py "$TR" 10 > "$OUT/step_breakdown_b1.txt" 2>&1
rm -rf "$OUT/stats"
cat "$OUT"/clock_fwd1.txt
