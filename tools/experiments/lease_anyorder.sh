#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-anyorder}; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
hipcc -O3 --offload-arch=gfx950 -o /tmp/anyorder tools/experiments/anyorder.hip > "$OUT/build.log" 2>&1 || { cat "$OUT/build.log"; exit 1; }
timeout -k 10 60 /tmp/anyorder > "$OUT/anyorder.txt" 2>&1; echo "rc $?" >> "$OUT/anyorder.txt"
cat "$OUT/anyorder.txt"
