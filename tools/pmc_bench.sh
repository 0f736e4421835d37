#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of the bench step itself (separate rocprofv3 passes, --kernel-trace only): what the
# merged forward grid (conv_lstm_multi8_kernel), which exists only inside the step, moves per launch.
# usage: tools/pmc_bench.sh <outdir> [bench args...]
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
