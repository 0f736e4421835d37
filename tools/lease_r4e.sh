#!/bin/bash
# Round-4 lease E: stencil kernel (opt-in: tile_rows = 1) tests + timing (two schedules of its scalar weight loads), the fold
# kernel with deeper unroll, the whole suite, a bench line + kernel trace for profiles/.
TAG=${1:-r4e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_stencil.py -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "passed|failed|Error|assert" "$OUT/new_tests.log" | tail -10
if [ $rc -eq 0 ]; then
for dt in bf16 f32; do for lib in nasa-niswan_amd/libnint_hip.so nasa-niswan_amd/build/libnint_stfence0.so; do for rows in 1 8; do
  echo "== configs[0] layer (4 -> 8, 3x3), full 100x154 grid, B=8, $dt, tile_rows=$rows (1: stencil, 8: implicit GEMM), $lib"
  timeout -k 10 200 python tools/kbench.py --hidden 8 --ks 3 --C 4 --dtype $dt --iters 200 --tile-rows $rows --lib $lib --only fwd0 2>&1 | grep fwd0 || exit 1
done; done; done | tee "$OUT/kbench_stencil.txt"
fi
bash tools/lease_tests_bench.sh $TAG/full
