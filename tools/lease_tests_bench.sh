#!/bin/bash
# One GPU lease: the -m gpu suite, the bench line, and a kernel trace of the bench with its per-step breakdown.
# usage (on the GPU box): bash tools/lease_tests_bench.sh <tag> [pytest args...]
TAG=${1:-lease}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -5 "$OUT/tests.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 > "$OUT/bench_line.json"
cut -c1-400 "$OUT/bench_line.json"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 > "$OUT/stats.log" 2>&1 )
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_steps10_kernel_stats.csv" \;
TR=$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/gaps.py "$TR" 10 > "$OUT/step_breakdown.txt" 2>&1
rm -rf "$OUT/stats"
cat "$OUT/step_breakdown.txt" | head -30
