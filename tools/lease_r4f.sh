#!/bin/bash
# Round-4 lease F: persistent tile loop in the gate / dgrad kernels (next tile's halo fill under this tile's epilogue):
# the whole suite first (bit-identity tests), then the bench A/B against a build with one tile per workgroup.
TAG=${1:-r4f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"
tail -6 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do for l in nasa-niswan_amd/build/libnint_tpw1.so -; do
  if [ "$l" = "-" ]; then L=""; else L="--lib $l"; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --long-steps 0 $L 2>> "$OUT/bench.err" | tail -1 > "$OUT/b.json" || exit 1
  python - "$OUT/b.json" "$l" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"lib={sys.argv[2]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step ", {k: v["us_per_launch"] for k, v in d["phases"]["per_step_us"].items() if not k.startswith(("wgrad", "fold", "pointwise1", "pointwise2", "dgrad2"))})
PY
done; done | tee "$OUT/ab.txt"
for b in 1 2 4; do for l in nasa-niswan_amd/build/libnint_tpw1.so -; do
  if [ "$l" = "-" ]; then L=""; else L="--lib $l"; fi
  timeout -k 10 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 $L 2>> "$OUT/bench.err" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b lib=$l', d['value'], d['ms_per_step'])" || exit 1
done; done | tee "$OUT/ab_small.txt"
