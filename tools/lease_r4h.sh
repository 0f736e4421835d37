#!/bin/bash
# Round-4 lease H: the suite with the forward wavefront on at B = 8 by default, then three fresh-process bench lines.
TAG=${1:-r4h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>> "$OUT/bench.err" | tail -1 > "$OUT/bench_$rep.json" || exit 1
  python - "$OUT/bench_$rep.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  200 steps: {d['value_200steps']:.1f}  wave={d['config']['wave']} roofline frac {d['roofline']['frac']}")
PY
done | tee "$OUT/bench_repeats.txt"
