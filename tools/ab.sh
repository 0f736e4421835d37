#!/bin/bash
# A/B bench lines on ONE device: tools/ab.sh "<bench args>" lib1.so lib2.so ... (each twice, interleaved); "-" = the product library
ARGS=$1; shift
for rep in 1 2; do for l in "$@"; do
  if [ "$l" = "-" ]; then L=""; else L="--lib $l"; fi
  python bench.py --no-cpu-baseline --no-kernel-rooflines --long-steps 0 $ARGS $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$l', d['config']['batch_per_gpu'], d['value'], d['ms_per_step'])"
done; done
