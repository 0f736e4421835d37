#!/bin/bash
# Round-4 lease A: the new parity tests first, then the whole -m gpu suite, the cfg3 A/B of the 8-wave kernel family
# (the decision the round-3 verdict asked for), and a baseline bench line + step breakdown.
TAG=${1:-r4a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_bias_grad.py "tests/test_gpu_bench_path.py::test_fused_trainer_step_at_the_bench_batch_of_8_vs_oracle" -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "layer|grad\.|loss|passed|failed|Error|assert" "$OUT/new_tests.log" | tail -70
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
for w in 1 2 1 2; do
  timeout -k 10 300 python bench.py --workload cfg3-1deg-hidden128 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline --long-steps 0 --wide $w 2>> "$OUT/bench.err" | tail -1 > "$OUT/cfg3_wide$w.json" || exit 1
  python - "$OUT/cfg3_wide$w.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"cfg3 B=2 wide={sys.argv[2]}: {d['value']:.2f} samples/s  {d['ms_per_step']:.2f} ms/step  gate0 in-step {r['ms_per_launch']*1e3:.1f} us frac {r['frac']:.3f}  wgrad0 {d['roofline_kernels'][1]['ms_per_launch']:.3f} ms frac {d['roofline_kernels'][1]['frac']:.3f}")
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err2" | tail -1 > "$OUT/bench_line.json"
python - "$OUT/bench_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["value_200steps"], {k: v["us_per_step"] for k, v in d["phases"]["per_step_us"].items()})
PY
