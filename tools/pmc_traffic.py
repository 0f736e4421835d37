#!/usr/bin/env python3
"""profiles/pmc_traffic.json from tools/pmc.sh / tools/pmc_bench.sh summaries: HBM bytes per launch of the kernels bench.py prices.
bytes = 2 * FETCH_SIZE_KB * 1024 (gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced streaming read,
MI355X_MICROARCH.md, HBM) + WRITE_SIZE_KB * 1024.  A kernel that appears with several grid sizes (the merged forward grid: 3 or 2
layers per launch) is averaged over all its launches, like its duration in a --kernel-trace --stats summary.
    usage: tools/pmc_traffic.py <source tag> <pmc_summary.txt> [<more summaries>...] > profiles/pmc_traffic.json"""
import json
import re
import sys

KEYS = {   # kernel-name prefix in the summary -> (bench key, launches of that kernel per priced "launch")
    "void conv_igemm_kernel<1, 0, 4, 1, 4, 8>": ("conv_igemm_fwd_layer0", 1),
    "void conv_lstm_multi8_kernel<1>": ("conv_lstm_multi8_fwd_wavefront", 1),   # the merged forward grid (nint_seq.wave = 2; bench passes)
    "void conv_igemm_kernel<1, 1, 1, 4, 4, 8>": ("conv_igemm_dgrad_layer0", 1),
    "void wgrad_wide_kernel<5, 2>": ("wgrad_layer0", 2),                 # x part + h part (round 4: the 8-wave 128-column kernel)
    "void lstm_bwd_pointwise_kernel<1>": ("lstm_bwd_pointwise_layer0", 1),
}
out = {"_comment": "HBM traffic per launch from rocprofv3 --pmc passes (tools/pmc.sh over tools/kbench.py, tools/pmc_bench.sh over bench.py; "
                   "FETCH_SIZE and WRITE_SIZE in separate passes): bytes = 2*FETCH_SIZE_KB*1024 (gfx950 FETCH_SIZE reads half of a wide "
                   "coalesced stream, MI355X_MICROARCH.md) + WRITE_SIZE_KB*1024.  Source: " + sys.argv[1]}
acc = {}
for path in sys.argv[2:]:
    for blk in open(path).read().split("== ")[1:]:
        name = blk.split("  grid=")[0]
        n = int(re.search(r"n=(\d+)", blk).group(1)) if re.search(r"n=(\d+)", blk) else 1
        for pre, (key, mult) in KEYS.items():
            if name.startswith(pre) and f"cfg1-20level/bf16/B8/{key}" not in out:
                f = re.search(r"FETCH_SIZE\s+([\d.]+)", blk)
                w = re.search(r"WRITE_SIZE\s+([\d.]+)", blk)
                if f and w:
                    if (path, key) in acc and not key.startswith("conv_lstm_multi8"):
                        continue                       # (the first block = the largest total = the layer-0 use of the instantiation)
                    a = acc.setdefault((path, key), [0, 0.0, 0.0, mult])
                    a[0] += n; a[1] += n * float(f.group(1)); a[2] += n * float(w.group(1))
    for (pth, key), (n, fs, wsz, mult) in list(acc.items()):
        if pth == path and f"cfg1-20level/bf16/B8/{key}" not in out and n:
            fk, wk = fs / n, wsz / n
            # (the stand-alone kbench passes list a layer-0 kernel first -- largest total -- and other layers' uses of the same
            # instantiation after it: only the merged grid is averaged over its grid sizes)
            out[f"cfg1-20level/bf16/B8/{key}"] = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "launches": mult,
                                                  "bytes_per_launch": int(mult * (2 * fk + wk) * 1024)}
print(json.dumps(out, indent=1))
