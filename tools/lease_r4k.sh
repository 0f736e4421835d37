#!/bin/bash
# Round-4 lease K: the exchange in two pieces (bit-equality test over two ranks on one device; a 1-rank RCCL bench line with and
# without it), wave = 2 against the default at B = 4 and B = 2, then the whole suite.
TAG=${1:-r4k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest "tests/test_gpu_train.py::test_two_ranks_with_the_exchange_in_two_pieces_end_on_the_same_bits" "tests/test_gpu_train.py::test_two_ranks_on_one_device_match_the_global_batch" -x -q -s -m gpu > "$OUT/new_tests.log" 2>&1; rc=$?
echo "pytest rc $rc" >> "$OUT/new_tests.log"; tail -5 "$OUT/new_tests.log"
for ov in "" "--overlap-allreduce" "" "--overlap-allreduce"; do
  timeout -k 10 300 python bench.py --force-dist $ov --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 2>> "$OUT/bench.err" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dist $ov', d['value'], d['ms_per_step'], 'loss', d['final_loss'], 'allreduce_ms', d['allreduce_ms'])" || exit 1
done | tee "$OUT/overlap_1rank.txt"
for b in 4 2; do for rep in 1 2 3; do for w in -1 2; do
  timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 --wave $w 2>> "$OUT/bench.err" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b wave=$w', d['value'], d['ms_per_step'], d['config']['wave'])" || exit 1
done; done; done | tee "$OUT/wave2_small.txt"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
