#!/bin/bash
# Step breakdown (rocprofv3 kernel trace -> tools/gaps.py) at a given batch size.  usage: tools/lease_b1prof.sh <tag> <batch> [bench args]
TAG=${1:-b1}; B=${2:-1}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 "$@" > "$OUT/stats.log" 2>&1 )
TR=$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/gaps.py "$TR" 10 > "$OUT/step_breakdown_b$B.txt" 2>&1
rm -rf "$OUT/stats"
cat "$OUT/step_breakdown_b$B.txt"
