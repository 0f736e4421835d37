#!/bin/bash
# lease GG: a long random-shape screen on the final code (all modes of tools/fuzz_shapes.py)
TAG=${1:-r4gg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout -k 10 1000 python tools/fuzz_shapes.py "$@" > "$OUT/fuzz_$name.log" 2>&1; echo "rc $?" >> "$OUT/fuzz_$name.log"; echo "$name: $(grep -c '^ok' "$OUT/fuzz_$name.log") ok, $(grep -c -i '^fail' "$OUT/fuzz_$name.log") failed; $(grep 'shapes ok\|ok;' "$OUT/fuzz_$name.log" | tail -1)"; }
run plain --n 350 --seed 601
run plain2 --n 200 --seed 607





