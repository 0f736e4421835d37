#!/bin/bash
# lease AA: BPTT grid 2 (fused top-layer step + bottom pointwise backward) with its two problems alternating in groups of 8 workgroups
# (-DNINT_PW_INTERLEAVE=1 build) against the fused step's workgroups first
TAG=${1:-r4aa}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for b in 8 2 4; do for rep in 1 2 3; do for lib in product nasa-niswan_amd/build/libnint_pwil.so; do
  L=""; [ $lib != product ] && L="--lib $lib"
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 $L 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b', '$lib'[-14:], d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'], 'loss', d['final_loss'])" || exit 1
done; done; done | tee "$OUT/pw_interleave_ab.txt"
