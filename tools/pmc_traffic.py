#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc.sh summary: HBM bytes per launch of the four kernels bench.py prices.
bytes = 2 * FETCH_SIZE_KB * 1024 (gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced streaming read,
MI355X_MICROARCH.md, HBM) + WRITE_SIZE_KB * 1024.    usage: tools/pmc_traffic.py <pmc_summary.txt> <source tag> > profiles/pmc_traffic.json"""
import json
import re
import sys

KEYS = {   # kernel-name prefix in the summary -> (bench key, launches of that kernel per priced "launch")
    "void conv_igemm_kernel<1, 0, 4, 1, 4, 8>": ("conv_igemm_fwd_layer0", 1),
    "void conv_igemm_kernel<1, 1, 1, 4, 4, 8>": ("conv_igemm_dgrad_layer0", 1),
    "void wgrad_wide_kernel<5, 2>": ("wgrad_layer0", 2),                 # x part + h part (round 4: the 8-wave 128-column kernel)
    "void lstm_bwd_pointwise_kernel<1>": ("lstm_bwd_pointwise_layer0", 1),
}
txt = open(sys.argv[1]).read()
out = {"_comment": "HBM traffic per launch from rocprofv3 --pmc passes (tools/pmc.sh over tools/kbench.py, FETCH_SIZE and WRITE_SIZE in "
                   "separate passes): bytes = 2*FETCH_SIZE_KB*1024 (gfx950 FETCH_SIZE reads half of a wide coalesced stream, "
                   "MI355X_MICROARCH.md) + WRITE_SIZE_KB*1024.  Source: " + sys.argv[2]}
for blk in txt.split("== ")[1:]:
    name = blk.split("  grid=")[0]
    for pre, (key, mult) in KEYS.items():
        if name.startswith(pre):
            f = re.search(r"FETCH_SIZE\s+([\d.]+)", blk)
            w = re.search(r"WRITE_SIZE\s+([\d.]+)", blk)
            if f and w:
                fk, wk = float(f.group(1)), float(w.group(1))
                out[f"cfg1-20level/bf16/B8/{key}"] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches": mult,
                                                      "bytes_per_launch": int(mult * (2 * fk + wk) * 1024)}
print(json.dumps(out, indent=1))
