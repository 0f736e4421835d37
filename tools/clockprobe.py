#!/usr/bin/env python3
"""Phase timeline and in-kernel clock of the gate / dgrad kernel (diagnostic build only).

    hipcc ... -DNINT_STAMP -> nasa-niswan_amd/build/libnint_stamp.so   (see tools/build_stamp.sh)
    python tools/clockprobe.py [--kernel fwd0|dgrad0|fwd1|dgrad1] [--lib nasa-niswan_amd/build/libnint_stamp.so]

Every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at entry, after the halo
fill, after the K loop and at exit (MI355X_MICROARCH.md, DVFS give-back item 6).  Prints the clock the
chip holds inside the K loop, the phase durations and the MFMA-issue occupancy of the loop."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nasa_niswan_amd as pkg  # noqa: E402
from nasa_niswan_amd.engine import LayerCfg, SeqEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="fwd0")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nasa-niswan_amd", "build", "libnint_stamp.so"))
    args = ap.parse_args()
    lib = pkg.load_library(args.lib)          # the -DNINT_STAMP build, by explicit path (tools/build_stamp.sh)
    rd = lib.nint_debug_read_stamps          # AttributeError: not a -DNINT_STAMP build
    rd.restype, rd.argtypes = C.c_int, [C.c_void_p, C.c_int]
    hidden, ks = (64, 32, 16), (5, 3, 3)
    cfgs, cin = [], 62
    for ch, k in zip(hidden, ks):
        cfgs.append(LayerCfg(cin, ch, k)); cin = ch
    eng = SeqEngine(cfgs, "bf16", "cuda")
    B, T, H, W = args.batch, 2, 100, 154
    ws = eng.acquire(B, T, H, W, True, False)
    eng.pack_weights([torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda") * 0.05 for c in cfgs],
                     [torch.zeros(4 * c.Ch, device="cuda") for c in cfgs])
    eng.forward(ws, torch.randn(B, T, 62, H, W, device="cuda"))
    for l in range(3):
        ws.dG[l].view(torch.bfloat16).normal_(std=0.05)
    l = int(args.kernel[-1])
    ly, g, es = eng.layers[l], C.byref(ws.g), eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    xs = ws.xs.data_ptr() + B * halo_px * ly.Cxp * es if l == 0 else ws.h[l - 1].data_ptr() + 2 * B * halo_px * ly.Cxp * es
    hs, cs = B * halo_px * ly.Chp * es, B * comp_px * ly.Chp * 4
    gs, dgs = B * comp_px * 4 * ly.Ch16 * es, B * halo_px * 4 * ly.Ch16 * es
    if args.kernel.startswith("fwd"):
        def fn():
            assert lib.nint_cell_fwd(C.byref(ly), g, eng.dt, B, C.c_void_p(xs), C.c_void_p(ws.h[l].data_ptr() + hs),
                                     C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.h[l].data_ptr() + 2 * hs),
                                     C.c_void_p(ws.c[l].data_ptr() + 2 * cs), C.c_void_p(ws.gates[l].data_ptr() + gs), None) == 0
        nsteps = (ly.Cxp + ly.Chp) // 32 * ly.k ** 2
        mfma_per_step = None
    elif args.kernel.startswith("fused"):
        dx = ws.dh[l - 1].data_ptr() if l > 0 else None
        def fn():
            assert lib.nint_cell_bwd_fused(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs), C.c_void_p(dx) if dx else None,
                                           C.c_void_p(ws.gates[l].data_ptr()), C.c_void_p(ws.c[l].data_ptr()),
                                           C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.dh[l].data_ptr()) if l < 2 else None,
                                           C.c_void_p(ws.dc[l].data_ptr()), C.c_void_p(ws.dG[l].data_ptr()), None) == 0
    else:
        dx = ws.dh[l - 1].data_ptr() if l > 0 else None
        def fn():
            assert lib.nint_conv_dgrad(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs),
                                       C.c_void_p(dx) if dx else None, C.c_void_p(ws.dh[l].data_ptr()), None) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    n = 0
    while True:
        for _ in range(50):
            fn()
        n += 50
        e1.record(); e1.synchronize()
        if e0.elapsed_time(e1) > args.seconds * 1e3:
            break
    avg_us = e0.elapsed_time(e1) / n * 1e3
    nw = 4096
    buf = np.zeros(nw * 8, dtype=np.uint64)
    assert rd(buf.ctypes.data, nw) == 0
    st = buf.reshape(nw, 4, 2).astype(np.int64)
    used = st[:, 3, 1] > 0
    st = st[used]
    st = st[st[:, 0, 1] >= st[:, 0, 1].max() - 100000]      # only the last launch (stale rows: earlier, larger grids)
    mt, rt = st[:, :, 0], st[:, :, 1]
    t0 = rt[:, 0].min()
    ph = lambda a, b: (rt[:, b] - rt[:, a]) / 100.0          # microseconds (100 MHz)
    clk = lambda a, b: (mt[:, b] - mt[:, a]) / np.maximum(rt[:, b] - rt[:, a], 1) * 0.1   # GHz
    print(f"{args.kernel}: {avg_us:.1f} us per launch over {n} back-to-back launches; {len(st)} workgroups stamped")
    for name, a, b in (("fill", 0, 1), ("K loop", 1, 2), ("epilogue", 2, 3), ("whole workgroup", 0, 3)):
        d, c = ph(a, b), clk(a, b)
        print(f"  {name:16s} median {np.median(d):7.2f} us  (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})   clock {np.median(c):.2f} GHz")
    kcyc = np.median(mt[:, 2] - mt[:, 1])
    print(f"  K loop shader cycles (median) {kcyc:.0f}")
    start = (rt[:, 0] - t0) / 100.0
    end = (rt[:, 3] - t0) / 100.0
    hist, edges = np.histogram(start, bins=12)
    print("  workgroup start times (us from first):", " ".join(f"{e:.0f}:{h}" for h, e in zip(hist, edges)))
    print(f"  last workgroup ends at {end.max():.1f} us")
    # concurrency of the phases over time: how many stamped workgroups are inside their K loop
    grid = np.linspace(0, end.max(), 25)
    k0, k1 = (rt[:, 1] - t0) / 100.0, (rt[:, 2] - t0) / 100.0
    print("  workgroups inside K loop at t:", " ".join(f"{int(((k0 <= t) & (k1 > t)).sum())}" for t in grid))
    if hasattr(lib, "nint_debug_read_hwid"):
        # which workgroups share a CU: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
        hw = np.zeros(nw * 2, dtype=np.uint32)
        lib.nint_debug_read_hwid.restype, lib.nint_debug_read_hwid.argtypes = C.c_int, [C.c_void_p, C.c_int]
        assert lib.nint_debug_read_hwid(hw.ctypes.data, nw) == 0
        hw = hw.reshape(nw, 2)[:len(used)][used]
        hw = hw[:len(st)] if len(hw) != len(st) else hw
        cu = ((hw[:, 1] & 15).astype(np.int64) << 8) | (((hw[:, 0] >> 13) & 7).astype(np.int64) << 5) | (((hw[:, 0] >> 12) & 1).astype(np.int64) << 4) | ((hw[:, 0] >> 8) & 15)
        order = np.argsort(start, kind="stable")
        first = order[:512] if len(order) >= 512 else order
        print("  distinct CUs seen:", len(set(cu.tolist())), " first 24 workgroups -> (xcc, se, sh, cu):",
              " ".join(f"{int(hw[b, 1] & 15)}.{int((hw[b, 0] >> 13) & 7)}.{int((hw[b, 0] >> 12) & 1)}.{int((hw[b, 0] >> 8) & 15)}" for b in range(24)))
        pairs = {}
        for b in first:
            pairs.setdefault(int(cu[b]), []).append(int(b))
        diffs = [p[1] - p[0] for p in pairs.values() if len(p) == 2]
        vals, cnts = np.unique(diffs, return_counts=True)
        print("  first-round workgroups sharing a CU: blockIdx difference histogram:", " ".join(f"{v}:{c}" for v, c in zip(vals, cnts)))
    print("  workgroups in fill/epilogue at t:", " ".join(f"{int((((start <= t) & (k0 > t)) | ((k1 <= t) & (end > t))).sum())}" for t in grid))


if __name__ == "__main__":
    main()
