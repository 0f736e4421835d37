#!/bin/bash
# the -m gpu suite only
TAG=${1:-suite}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?; echo "pytest rc $rc" >> "$OUT/tests.log"; tail -15 "$OUT/tests.log"
