#!/bin/bash
# Round-4 lease B: the 8-wave 128-column weight-gradient kernel -- its tests, per-kernel A/B at BASELINE configs[3] shapes, the
# configs[3] step, then the whole -m gpu suite.
TAG=${1:-r4b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_wgrad_families.py tests/test_gpu_bias_grad.py -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "wide=|layer [0-9]|passed|failed|Error|assert" "$OUT/new_tests.log" | tail -60
[ $rc -ne 0 ] && exit 1
for w in 1 2 1 2; do
  echo "== kbench cfg3 shapes, B=2 T=24, wide=$w"
  timeout -k 10 300 python tools/kbench.py --hidden 128,128,128 --ks 3,3,3 --H 190 --W 298 --batch 2 --T 24 --iters 5 --wide $w --only wgrad0,wgrad1 2>&1 | tail -3 || exit 1
done | tee "$OUT/kbench_cfg3.txt"
for w in 1 0 1 0; do
  timeout -k 10 300 python bench.py --workload cfg3-1deg-hidden128 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline --long-steps 0 --wide $w 2>> "$OUT/bench.err" | tail -1 > "$OUT/cfg3_wide$w.json" || exit 1
  python - "$OUT/cfg3_wide$w.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"cfg3 B=2 wide={sys.argv[2]}: {d['value']:.2f} samples/s  {d['ms_per_step']:.2f} ms/step  gate0 in-step {r['ms_per_launch']*1e3:.1f} us frac {r['frac']:.3f}  wgrad0 {d['roofline_kernels'][1]['ms_per_launch']:.3f} ms frac {d['roofline_kernels'][1]['frac']:.3f}", {k: v["us_per_step"] for k, v in d["phases"]["per_step_us"].items() if k.startswith(("wgrad", "fold"))})
PY
done | tee "$OUT/cfg3_ab.txt"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
