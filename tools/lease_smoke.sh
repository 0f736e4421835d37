#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/smoke
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke/smoke.log 2>&1; echo "rc $?" >> gpurun_out/smoke/smoke.log; tail -5 gpurun_out/smoke/smoke.log
