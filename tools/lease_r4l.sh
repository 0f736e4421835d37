#!/bin/bash
# Round-4 lease L: which merged-grid mode for small batches (B = 1, 2, 4: wave 1 / 2 / 3), and the 1-rank RCCL line with and
# without the two-piece exchange (what else, if anything, reaches stdout under --force-dist).
TAG=${1:-r4l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for ov in "" "--overlap-allreduce" "" "--overlap-allreduce"; do
  timeout -k 10 300 python bench.py --force-dist $ov --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 > "$OUT/fd.out" 2>> "$OUT/bench.err"
  echo "stdout lines: $(wc -l < "$OUT/fd.out"); first 120 chars of each:"; cut -c1-120 "$OUT/fd.out"
  tail -1 "$OUT/fd.out" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dist $ov', d['value'], d['ms_per_step'], 'loss', d['final_loss'], 'allreduce_ms', d['allreduce_ms'])"
done | tee "$OUT/overlap_1rank.txt"
for b in 1 2 4; do for rep in 1 2 3; do for w in 1 2 3; do
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b wave=$w', d['value'], d['ms_per_step'])" || exit 1
done; done; done | tee "$OUT/wave_small.txt"
