#!/bin/bash
# lease DD: non-temporal loads of read-once streams: the gate stash / dh in the pointwise backward (pwnt), the f32 input in the pack kernel (packnt)
TAG=${1:-r4dd}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for b in 8 2; do for rep in 1 2 3; do for lib in product nasa-niswan_amd/build/libnint_pwnt.so nasa-niswan_amd/build/libnint_packnt.so; do
  L=""; [ $lib != product ] && L="--lib $lib"
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 $L 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b', '$lib'[-16:], d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done; done | tee "$OUT/nt_ab.txt"
