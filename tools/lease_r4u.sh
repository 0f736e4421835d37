#!/bin/bash
TAG=${1:-r4u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_shapes.py tests/test_gpu_bench_path.py tests/test_gpu_fullsize.py -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -5 "$OUT/tests.log"
