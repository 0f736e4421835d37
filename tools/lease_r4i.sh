#!/bin/bash
# Round-4 lease I: the dense-K MFMA kernel for tiny layers (tests, full-grid launch time against the stencil kernel and the padded
# tiles), then an A/B of the forward merged grid with the narrow layers on 8-row tiles (wave code 4) against wave = 2.
TAG=${1:-r4i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_stencil.py tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "dense-K|passed|failed|Error|assert" "$OUT/new_tests.log" | tail -24
if [ $rc -eq 0 ]; then
for dt in bf16 f32; do for rows in 0 1 8 0 8; do
  echo "== configs[0] layer (4 -> 8, 3x3), full 100x154 grid, B=8, $dt, tile_rows=$rows (0: dense-K MFMA, 1: stencil, 8: padded implicit GEMM)"
  timeout -k 10 200 python tools/kbench.py --hidden 8 --ks 3 --C 4 --dtype $dt --iters 200 --tile-rows $rows --only fwd0 2>&1 | grep fwd0 || exit 1
done; done | tee "$OUT/kbench_tiny.txt"
fi
for rep in 1 2 3; do for w in 2 4; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --wave $w --phase-events 30 2>> "$OUT/bench.err" | tail -1 > "$OUT/wave${w}_$rep.json" || exit 1
  python - "$OUT/wave${w}_$rep.json" $w $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"wave={sys.argv[2]} process {sys.argv[3]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  phases {d['phase_ms']}")
PY
done; done | tee "$OUT/wave_rows8.txt"
