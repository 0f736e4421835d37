#!/bin/bash
# Lease for the wide-kernel bring-up: its tests, then A/B bench lines (family forced) on the same device.
TAG=${1:-wide}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_wide.py -x -q "$@" > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"
tail -15 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
for w in ${WIDES:-1 0 1 0}; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --long-steps 0 --wide $w 2>> "$OUT/bench.err" | tail -1 > "$OUT/bench_wide$w.json" || exit 1
  python - "$OUT/bench_wide$w.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"wide={sys.argv[2]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  gate0 in-step {r['ms_per_launch']*1e3:.1f} us (loop {r['ms_per_launch_loop']*1e3:.1f})  frac {r['frac']:.3f}  loss {d['final_loss']}")
PY
done
