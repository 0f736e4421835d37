#!/bin/bash
# Round-4 lease J: suite with the 8-row merged forward grid as the B = 8 default; A/B of the backward pair with 8-row fused tiles (wave 3).
TAG=${1:-r4j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3 4; do for w in 2 3; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --wave $w --phase-events 30 2>> "$OUT/bench.err" | tail -1 > "$OUT/wave${w}_$rep.json" || exit 1
  python - "$OUT/wave${w}_$rep.json" $w $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"wave={sys.argv[2]} process {sys.argv[3]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  phases {d['phase_ms']}")
PY
done; done | tee "$OUT/wave_bwd8.txt"
