#!/bin/bash
# Bench lines over batch sizes / storage types / workloads with the current code (one JSON line each).
# usage: tools/sweep.sh <out.jsonl>     (on the GPU box; ~2 minutes)
OUT=${1:-gpurun_out/sweep.jsonl}
: > "$OUT"
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 "$@" 2>/dev/null | tail -1 >> "$OUT"; }
run --batch 1
run --batch 2
run --batch 4
run --batch 8
run --batch 32
run --batch 8 --dtype f32
run --batch 8 --workload cfg1-refpinned
run --batch 8 --workload cfg4-multitracer-40lev
run --batch 2 --workload cfg3-1deg-hidden128 --steps 5 --warmup 2
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    c = d["config"]
    print(f'{c["workload"]:26s} B={c["batch_per_gpu"]:<3d} {d["dtype"]:5s} {d["value"]:9.1f} samples/s  {d["ms_per_step"]:9.3f} ms/step  '
          f'gate-kernel frac {d["roofline"]["frac"]:.3f}  executed-MFMA frac {d["whole_step_executed_mfma_frac"]:.3f}')
PY
