#!/bin/bash
# lease CC: split-K fold with non-temporal loads of the partial slabs (-DNINT_FOLD_NT build) against the product; in-step fold time from the probe pass
TAG=${1:-r4cc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for b in 8 1; do for rep in 1 2; do for lib in nasa-niswan_amd/build/libnint_foldnt.so nasa-niswan_amd/build/libnint_foldnt2.so; do
  L=""; L="--lib $lib"
  timeout -k 10 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --long-steps 0 $L 2>> "$OUT/bench.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['phases']['per_step_us']; print('B=$b', '$lib'[-15:], d['value'], d['ms_per_step'], 'fold', p['fold0']['us_per_step'], 'wgrad0', p['wgrad0']['us_per_step'])" || exit 1
done; done; done | tee "$OUT/fold_nt_ab.txt"
