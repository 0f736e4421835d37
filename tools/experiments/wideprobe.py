#!/usr/bin/env python3
"""Where the K-steps of the 8-wave LDS-weight gate kernel (csrc/conv_wide.hip) spend their cycles (diagnostic build only).

    bash tools/build_stamp.sh && python tools/wideprobe.py [--batch 8] [--layer 0]

Every wave accumulates s_memtime differences over its whole run: P1 (DMA issue + fragment reads), wait at the first
barrier, MFMA issue, counted vmcnt wait, wait at the second barrier, epilogues, prologue.  Printed per wave group
(waves 0-3 / 4-7: the two waves of a SIMD) as cycles per K-step, with the in-kernel clock."""
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--wide", type=int, default=2, help="nint_layer.wide: 2 = the library's tile size, 3 / 4 = 256- / 512-pixel tiles")
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nasa-niswan_amd", "build", "libnint_stamp.so"))
    args = ap.parse_args()
    lib = pkg.load_library(args.lib)
    rd = lib.nint_debug_read_wide_stamps
    rd.restype, rd.argtypes = C.c_int, [C.c_void_p, C.c_int]
    cfgs = [LayerCfg(62, 64, 5)]
    from nasa_niswan_amd import engine
    engine.FORCE_WIDE = args.wide
    eng = SeqEngine(cfgs, "bf16", "cuda")
    B, T, H, W = args.batch, 2, 100, 154
    ws = eng.acquire(B, T, H, W, True, False)
    eng.pack_weights([torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda") * 0.05 for c in cfgs],
                     [torch.zeros(4 * c.Ch, device="cuda") for c in cfgs])
    eng.forward(ws, torch.randn(B, T, 62, H, W, device="cuda"))
    ly, g, es = eng.layers[0], C.byref(ws.g), eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    xs = ws.xs.data_ptr() + B * halo_px * ly.Cxp * es
    hs, cs, gs = B * halo_px * ly.Chp * es, B * comp_px * ly.Chp * 4, B * comp_px * 4 * ly.Ch16 * es

    def fn():
        assert lib.nint_cell_fwd(C.byref(ly), g, eng.dt, B, C.c_void_p(xs), C.c_void_p(ws.h[0].data_ptr() + hs),
                                 C.c_void_p(ws.c[0].data_ptr() + cs), C.c_void_p(ws.h[0].data_ptr() + 2 * hs),
                                 C.c_void_p(ws.c[0].data_ptr() + 2 * cs), C.c_void_p(ws.gates[0].data_ptr() + gs), None) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record(); e1.synchronize()
    print(f"{e0.elapsed_time(e1) / args.iters * 1e3:.1f} us per launch over {args.iters} back-to-back launches")
    nw = 256
    buf = np.zeros(nw * 8 * 16, dtype=np.uint64)
    assert rd(buf.ctypes.data, nw) == 0
    st = buf.reshape(nw, 8, 16).astype(np.int64)
    st = st[st[:, 0, 10] > 0]
    S = st[:, :, 10].astype(float)
    names = ["P1 (DMA issue, ds_reads)", "wait at barrier 1", "MFMA issue", "vmcnt wait", "wait at barrier 2", "epilogues + tile switch", "prologue"]
    clk = np.median(st[:, :, 8] / np.maximum(st[:, :, 9], 1)) * 0.1
    print(f"{len(st)} workgroups, {int(np.median(S))} K-steps per wave, in-kernel clock {clk:.2f} GHz, "
          f"wave lifetime {np.median(st[:, :, 9]) / 100:.1f} us (p10 {np.percentile(st[:, :, 9], 10) / 100:.1f}, p90 {np.percentile(st[:, :, 9], 90) / 100:.1f})")
    t0 = st[:, 0, 12].min()
    print("workgroup start (us after the first):", " ".join(f"{v:.0f}" for v in np.percentile((st[:, 0, 12] - t0) / 100.0, [0, 25, 50, 75, 100])))
    light = int(st[:, :, 0].sum()) == 0
    if light:     # -DNINT_STAMP=1 build: whole-run totals only, the K loop is undisturbed
        loop = (st[:, :, 8] - st[:, :, 6] - st[:, :, 5]) / S
        print(f"K loop: {np.median(loop):.0f} shader cycles per K-step (p10 {np.percentile(loop, 10):.0f}, p90 {np.percentile(loop, 90):.0f}); "
              f"epilogue + tile switch {np.median(st[:, :, 5] / np.maximum(st[:, :, 11], 1)):.0f} cycles per tile; prologue {np.median(st[:, :, 6]):.0f}")
        return
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(grp)
        for i, n in enumerate(names):
            v = st[:, sl, i]
            if i < 5:
                print(f"   {n:28s} {np.median(v / S[:, sl]):8.0f} cycles per K-step")
            else:
                print(f"   {n:28s} {np.median(v):8.0f} cycles per run ({np.median(v / np.maximum(st[:, sl, 11], 1)):.0f} per tile)")
        for i, n in ((13, "   P1: weights + chunk issue"), (14, "   P1: fragment reads issue (+ last-step c_prev loads)")):
            print(f"   {n:50s} {np.median(st[:, sl, i] / S[:, sl]):8.0f} cycles per K-step (of P1)")


if __name__ == "__main__":
    main()
