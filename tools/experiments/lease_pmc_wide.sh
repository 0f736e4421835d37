#!/bin/bash
# PMC passes of the layer-0 gate launch, 4-wave kernel (wide=1) next to the 8-wave LDS-weight kernel (wide=3: 256-pixel tiles)
TAG=${1:-pmcw}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for w in 1 3; do
  bash tools/pmc.sh "$OUT/sq_w$w" --iters 3 --only fwd0 --wide $w > "$OUT/sq_w$w.log" 2>&1
  python tools/pmc_summary.py "$OUT/sq_w$w" > "$OUT/sq_w$w.txt" 2>&1
  bash tools/pmc2.sh "$OUT/mem_w$w" --iters 3 --only fwd0 --wide $w > "$OUT/mem_w$w.log" 2>&1
  python tools/pmc_summary.py "$OUT/mem_w$w" > "$OUT/mem_w$w.txt" 2>&1
  rm -rf "$OUT"/sq_w$w/pass*/*/*.db "$OUT"/mem_w$w/pass*/*/*.db
done
tail -30 "$OUT"/sq_w3.txt
