#!/bin/bash
# lease Y: B = 1: wave 1 (forward wavefront on the layers' own tiles + bottom dgrad with the fused step) against 5 (the same forward + the two BPTT pairs of wave 4)
TAG=${1:-r4y}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/wave_bits.py 1 5 --dtype f32 --batch 1 > "$OUT/bits_f32.txt" 2>&1 || { cat "$OUT/bits_f32.txt"; exit 1; }
tail -1 "$OUT/bits_f32.txt"
timeout -k 10 300 python tools/wave_bits.py 1 5 --batch 1 > "$OUT/bits_bf16.txt" 2>&1 || { cat "$OUT/bits_bf16.txt"; exit 1; }
tail -1 "$OUT/bits_bf16.txt"
for b in 1 2; do for rep in 1 2 3; do for w in 1 5 4; do
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b wave=$w', d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done; done | tee "$OUT/wave5_ab.txt"
