#!/bin/bash
# the driver's command on a fresh box: python bench.py (defaults), one JSON line
TAG=${1:-bl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench_line.json"
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print('$TAG', d['value'], d['ms_per_step'], d['value_200steps'], d['roofline']['frac'], d['cpu_baseline']['value'])"
