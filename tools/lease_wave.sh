#!/bin/bash
# Small-batch A/B of the forward wavefront (bench.py --wave 0|1) + the suites that cover the touched kernels.
TAG=${1:-wave}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_small_kernels.py tests/test_gpu_shapes.py tests/test_gpu_train.py -x -q -m gpu > "$OUT/tests.log" 2>&1 || { tail -30 "$OUT/tests.log"; exit 1; }
tail -2 "$OUT/tests.log"
for b in 1 2 4 8; do for w in 0 1 0 1; do
  timeout -k 10 200 python bench.py --batch $b --wave $w --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b wave=$w', d['value'], d['ms_per_step'])" || exit 1
done; done | tee "$OUT/ab.txt"
