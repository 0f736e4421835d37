#!/bin/bash
# Round-4 lease D: the VALU stencil path (tests, full-grid gate launch time against the implicit-GEMM kernel on the same shape),
# then fresh-process repeats of the bench line with the weight-gradient family forced off / left to the library.
TAG=${1:-r4d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_stencil.py -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "stencil vs gemm|passed|failed|Error|assert" "$OUT/new_tests.log" | tail -40
if [ $rc -eq 0 ]; then
for dt in bf16 f32; do for rows in 0 8 0 8; do
  echo "== configs[0] layer (4 -> 8, 3x3) on the full 100x154 grid, B=8, $dt, tile_rows=$rows (0: stencil kernel, 8: implicit GEMM)"
  timeout -k 10 200 python tools/kbench.py --hidden 8 --ks 3 --C 4 --dtype $dt --iters 200 --tile-rows $rows --only fwd0 2>&1 | grep fwd0 || exit 1
done; done | tee "$OUT/kbench_stencil.txt"
fi
for rep in 1 2 3 4; do for w in 1 0; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 200 --wide $w 2>> "$OUT/bench.err" | tail -1 > "$OUT/cfg1_wide${w}_$rep.json" || exit 1
  python - "$OUT/cfg1_wide${w}_$rep.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"cfg1 B=8 wide={sys.argv[2]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  200 steps: {d['value_200steps']:.1f}")
PY
done; done | tee "$OUT/cfg1_repeats.txt"
