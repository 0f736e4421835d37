#!/bin/bash
# lease S: the dgrad pair at the smallest batches: wave 1 (forward wavefront on the layers' own tiles + bottom dgrad with the top
# layer's fused step), 4 (forward on 8-row tiles + dgrad pair), 5 (forward on the layers' own tiles + dgrad pair)
TAG=${1:-r4s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for b in 1 2; do for rep in 1 2 3; do for w in 1 4 5; do
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b wave=$w', d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done; done | tee "$OUT/wave5_ab.txt"
