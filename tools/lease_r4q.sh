#!/bin/bash
# lease Q: wave = 4 (forward wavefront + the bottom layer's dgrad of time u+1 in one grid with layer 1's dgrad of time u) against wave = 2
TAG=${1:-r4q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/wave_bits.py 2 4 --dtype f32 --batch 4 > "$OUT/bits_f32.txt" 2>&1 || { cat "$OUT/bits_f32.txt"; exit 1; }
tail -2 "$OUT/bits_f32.txt"
timeout -k 10 300 python tools/wave_bits.py 2 4 > "$OUT/bits_bf16.txt" 2>&1 || { cat "$OUT/bits_bf16.txt"; exit 1; }
tail -13 "$OUT/bits_bf16.txt"
for rep in 1 2 3; do for w in 2 4; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 --wave $w 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=8 wave=$w', d['value'], d['ms_per_step'], d.get('phase_ms'), 'loss', d['final_loss'])" || exit 1
done; done | tee "$OUT/wave4_ab.txt"
