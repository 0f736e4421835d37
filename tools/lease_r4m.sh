#!/bin/bash
# Round-4 lease M: the suite on the final code, then the round's final profile set.
TAG=${1:-r4m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"
tail -4 "$OUT/tests.log"
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/profile_round.sh $TAG/prof
