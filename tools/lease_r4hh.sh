#!/bin/bash
# lease HH: replay case #125 of seed 601 (head weight gradient 5.5e-2 off in bf16) under the launch schedules 0 / 2 / 4 / 1
TAG=${1:-r4hh}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for w in 0 2 4 1; do
  echo "== wave $w"; timeout -k 10 300 python tools/fuzz_shapes.py --n 200 --seed 601 --only 125 --force-wave $w 2>&1 | grep -v amdgpu.ids
done > "$OUT/replay125.txt" 2>&1
cat "$OUT/replay125.txt" | cut -c1-600
