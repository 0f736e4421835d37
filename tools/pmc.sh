#!/bin/bash
# Collect PMC counters for the per-kernel micro-benchmark, one rocprofv3 pass per counter group
# (PMC runs carry --kernel-trace only).  Every pass runs under its own timeout and a failed pass does not stop the
# others (an over-subscribed counter group makes the aborted profiler hang).  usage: tools/pmc.sh <outdir> [kbench args...]
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 $R/tools/kbench.py "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
