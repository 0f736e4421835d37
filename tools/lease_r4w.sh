#!/bin/bash
# Round-4 final lease: the suite, the bench line, the kernel trace + step breakdown, PMC traffic of the bench step, the sweep.
TAG=${1:-r4w}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"; tail -3 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
bash tools/profile_round.sh $TAG/prof
bash tools/pmc_bench.sh "$OUT/pmcb" > "$OUT/pmcb.log" 2>&1
python tools/pmc_summary.py "$OUT/pmcb" > "$OUT/pmc_bench_summary.txt" 2>&1
rm -rf "$OUT/pmcb"/pass*/*/*.db 2>/dev/null
grep -A3 "conv_lstm_multi8" "$OUT/pmc_bench_summary.txt" | head -12
cat "$OUT/prof/sweep.txt"
