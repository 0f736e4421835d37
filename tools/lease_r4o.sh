#!/bin/bash
# lease O: which BPTT launches gain from running side by side (two streams, bench shape)
out=gpurun_out/${1:-r4o}; mkdir -p $out
timeout -k 10 300 python tools/overlap_bwd_probe.py > $out/overlap_bwd.txt 2>&1; echo "rc $?" >> $out/overlap_bwd.txt
timeout -k 10 200 python tools/overlap_probe.py > $out/overlap_fwd.txt 2>&1; echo "rc $?" >> $out/overlap_fwd.txt
cat $out/overlap_bwd.txt $out/overlap_fwd.txt
