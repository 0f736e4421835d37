#!/usr/bin/env python3
"""Register / spill table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/regs.py nasa-niswan_amd/csrc/conv_igemm.hip [name filter] [-DNINT_EXPERIMENT ...]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", f"-I{root}/include",
       f"-I{root}/nasa-niswan_amd/csrc", "-c", src, "-o", "/tmp/regs_tmp.o", "-Rpass-analysis=kernel-resource-usage"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for blk in txt.split("remark: Function Name: ")[1:]:
    name = blk.split(" [-Rpass")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    g = lambda k: re.search(k + r": (\d+)", blk).group(1)
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{dem[:70]:70s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {scratch:>4s} occ {occ}")
