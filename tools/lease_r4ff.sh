#!/bin/bash
# lease FF: the pointwise pass inside BPTT grid 2 with two items per thread and loop turn (product) against one (-DNINT_PW_U=1)
TAG=${1:-r4ff}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_shapes.py -m gpu -x -q -k "dgrad_pair" > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"; tail -2 "$OUT/tests.log"
for b in 8 2 4; do for rep in 1 2 3; do for lib in nasa-niswan_amd/build/libnint_pwu1.so product; do
  L=""; [ $lib != product ] && L="--lib $lib"
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --phase-events 30 $L 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phase_ms']; print('B=$b', '$lib'[-15:], d['value'], d['ms_per_step'], 'fwd', p['pack_forward'], 'bwd', p['bptt_wgrad_fold'])" || exit 1
done; done; done | tee "$OUT/pw_u_ab.txt"
