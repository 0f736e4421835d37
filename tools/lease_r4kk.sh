#!/bin/bash
TAG=${1:-r4kk}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench_line.json"
python - "$OUT/bench_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["value_200steps"], d["config"]["wave"], d["phases"]["step_ms_with_probes"], d["phases"]["schedule_differs_from_timed_steps"])
for k, v in d["phases"]["per_step_us"].items(): print("  ", k, v)
for r in d["roofline_kernels"]: print(r["kernel"][:64], r["frac"], r["ms_per_launch"], r["timing"], r["launches_per_step"])
PY

