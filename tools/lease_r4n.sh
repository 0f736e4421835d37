#!/bin/bash
# Round-4 lease N: bench line with the merged forward grid as the headline roofline kernel; its HBM traffic by PMC passes over the
# bench step; the suite; the final profile set.
TAG=${1:-r4n}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_shapes.py -x -q -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/new_tests.log"; tail -3 "$OUT/new_tests.log"
[ $rc -ne 0 ] && exit 1
bash tools/pmc_bench.sh "$OUT/pmcb" > "$OUT/pmcb.log" 2>&1
python tools/pmc_summary.py "$OUT/pmcb" > "$OUT/pmc_bench_summary.txt" 2>&1
rm -rf "$OUT/pmcb"/pass*/*/*.db 2>/dev/null
grep -A3 "conv_lstm_multi8" "$OUT/pmc_bench_summary.txt" | head -20
bash tools/profile_round.sh $TAG/prof nosweep
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -3 "$OUT/tests.log"
