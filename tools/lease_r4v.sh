#!/bin/bash
# lease V: the random-shape screen with the launch schedules 0 / 1 / 2 / 4 / engine rule in rotation
TAG=${1:-r4v}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python tools/fuzz_shapes.py --n 100 --seed 501 > "$OUT/fuzz_plain.log" 2>&1; echo "rc $?" >> "$OUT/fuzz_plain.log"; tail -3 "$OUT/fuzz_plain.log"
timeout -k 10 400 python tools/fuzz_shapes.py --big --n 30 --seed 502 > "$OUT/fuzz_big.log" 2>&1; echo "rc $?" >> "$OUT/fuzz_big.log"; tail -3 "$OUT/fuzz_big.log"
timeout -k 10 300 python tools/fuzz_shapes.py --wide --n 30 --seed 503 > "$OUT/fuzz_wide.log" 2>&1; echo "rc $?" >> "$OUT/fuzz_wide.log"; tail -3 "$OUT/fuzz_wide.log"
timeout -k 10 300 python tools/fuzz_shapes.py --trainer --n 30 --seed 504 > "$OUT/fuzz_trainer.log" 2>&1; echo "rc $?" >> "$OUT/fuzz_trainer.log"; tail -3 "$OUT/fuzz_trainer.log"
