#!/bin/bash
# lease LL: the library before the probe change (sources of the previous commit, libnint_prev.so) against the product, alternating
TAG=${1:-r4ll}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for lib in product nasa-niswan_amd/build/libnint_prev.so product nasa-niswan_amd/build/libnint_prev.so; do
  L=""; [ $lib != product ] && L="--lib $lib"
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 $L 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=8', '$lib'[-15:], d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done | tee "$OUT/prev_ab.txt"
