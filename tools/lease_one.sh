#!/bin/bash
# one test file on the GPU box: bash tools/lease_one.sh <tag> <pytest args...>
TAG=${1:-one}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest "$@" -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "pytest rc $?" >> "$OUT/tests.log"; tail -12 "$OUT/tests.log"
