#!/bin/bash
# lease P: the merged forward grid with its problems ALTERNATING in groups of 8 workgroups (wave bit 4) against contiguous ranges
TAG=${1:-r4p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for w in 2 18; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=8 wave=$w', d['value'], d['ms_per_step'], d.get('phase_ms'))" || exit 1
done; done | tee -a "$OUT/interleave_ab.txt"
