#!/bin/bash
# Round-4 lease G: (1) fresh-process repeats of the bench step with the merged-grid wavefront forced on / off at B = 8, with the
# phases of the step bracketed by HIP events (which phase a slow process loses its time in); (2) the round's profile set.
TAG=${1:-r4g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4 5 6; do for w in 0 1; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --wave $w --phase-events 30 2>> "$OUT/bench.err" | tail -1 > "$OUT/wave${w}_$rep.json" || exit 1
  python - "$OUT/wave${w}_$rep.json" $w $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"wave={sys.argv[2]} process {sys.argv[3]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  phases {d['phase_ms']}")
PY
done; done | tee "$OUT/wave_repeats.txt"
bash tools/profile_round.sh $TAG/prof
