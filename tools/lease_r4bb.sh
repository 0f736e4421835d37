#!/bin/bash
# lease BB: larger batches: time-major (wave 0, the engine's rule above B = 8) against wave 4
TAG=${1:-r4bb}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for b in 16 32 12; do for rep in 1 2; do for w in 0 4 2; do
  timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 15 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b wave=$w', d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done; done | tee "$OUT/big_batches.txt"
