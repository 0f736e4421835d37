#!/bin/bash
# One round's profile set on the GPU box (writes under gpurun_out/<tag>/; copy what is to be judged into profiles/):
#   <tag>/bench_line.json            python bench.py --steps 20 --warmup 5 (un-profiled, with the CPU baseline)
#   <tag>/stats/...kernel_stats.csv  rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 3`
#   <tag>/pmc_summary.txt            tools/pmc.sh passes over tools/kbench.py (SQ_*, FETCH_SIZE, WRITE_SIZE in separate passes)
#   <tag>/sweep.jsonl                tools/sweep.sh
# usage: tools/profile_round.sh <tag> [nosweep]
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench_line.json"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 > "$OUT/stats.log" 2>&1 )
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_steps10_kernel_stats.csv" \;
TR=$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/gaps.py "$TR" 10 > "$OUT/step_breakdown.txt" 2>&1
rm -rf "$OUT/stats"
grep '^{"metric"' "$OUT/stats.log" | tail -1 > "$OUT/bench_line_profiled.json"
bash tools/pmc.sh "$OUT/pmc" --iters 2 --only fwd0,dgrad0,wgrad0,pointwise0,fwd1,dgrad1,fwd2 > "$OUT/pmc.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1
if [ "$2" != "nosweep" ]; then bash tools/sweep.sh "$OUT/sweep.jsonl" > "$OUT/sweep.txt" 2>&1; fi
rm -rf "$OUT/pmc"/pass*/*/*.db 2>/dev/null
du -sh "$OUT"
cat "$OUT/bench_line.json" | cut -c1-300
head -12 "$OUT/bench_steps10_kernel_stats.csv" | cut -c1-150
