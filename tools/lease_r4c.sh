#!/bin/bash
# Round-4 lease C: the 128-column weight-gradient kernel on 5x5 / 7x7 layers: tests, per-kernel A/B, the bench step A/B.
TAG=${1:-r4c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_wgrad_families.py -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; grep -E "wide=|passed|failed|Error|assert" "$OUT/new_tests.log" | tail -40
[ $rc -ne 0 ] && exit 1
for w in 1 2 1 2; do
  echo "== kbench bench workload (cfg1-20level), B=8 T=12, wide=$w"
  timeout -k 10 300 python tools/kbench.py --iters 10 --wide $w --only wgrad0,wgrad1,wgrad2 2>&1 | tail -4 || exit 1
done | tee "$OUT/kbench_cfg1.txt"
for l in nasa-niswan_amd/build/libnint_stream1.so nasa-niswan_amd/libnint_hip.so; do
  echo "== kbench cfg3 shapes, lib=$l"
  timeout -k 10 300 python tools/kbench.py --hidden 128,128,128 --ks 3,3,3 --H 190 --W 298 --batch 2 --T 24 --iters 5 --lib $l --only wgrad0,wgrad1 2>&1 | tail -3 || exit 1
done | tee "$OUT/kbench_cfg3_stream.txt"
for w in 1 0 1 0; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --long-steps 0 --wide $w 2>> "$OUT/bench.err" | tail -1 > "$OUT/cfg1_wide$w.json" || exit 1
  python - "$OUT/cfg1_wide$w.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"cfg1 B=8 wide={sys.argv[2]}: {d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  wgrad0 {d['roofline_kernels'][1]['ms_per_launch']:.3f} ms frac {d['roofline_kernels'][1]['frac']:.3f}", {k: v["us_per_step"] for k, v in d["phases"]["per_step_us"].items() if k.startswith(("wgrad", "fold"))})
PY
done | tee "$OUT/cfg1_ab.txt"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
