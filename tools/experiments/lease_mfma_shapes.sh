#!/bin/bash
# builds tools/experiments/mfma_shapes.hip on the GPU box and runs it on random and on all-zero operands
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfma}; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shapes tools/experiments/mfma_shapes.hip > "$OUT/build.log" 2>&1 || { cat "$OUT/build.log"; exit 1; }
timeout -k 10 120 /tmp/mfma_shapes 1 > "$OUT/mfma_shapes.txt" 2>&1
timeout -k 10 120 /tmp/mfma_shapes 0 >> "$OUT/mfma_shapes.txt" 2>&1
cat "$OUT/mfma_shapes.txt"
