#!/usr/bin/env python3
"""Do the gate kernels of different layers overlap when they are issued on different HIP streams?
The (t, layer) wavefront makes fwd0(t+2), fwd1(t+1), fwd2(t) mutually independent (likewise the dgrad kernels).
Measures back-to-back on one stream vs the three on three streams, in both issue orders.
    python tools/overlap_probe.py [--batch 8] [--iters 30]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nasa_niswan_amd as pkg  # noqa: E402
from nasa_niswan_amd.engine import LayerCfg, SeqEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    lib = pkg.load_library()
    hidden, ks = (64, 32, 16), (5, 3, 3)
    cfgs, cin = [], 62
    for ch, k in zip(hidden, ks):
        cfgs.append(LayerCfg(cin, ch, k)); cin = ch
    eng = SeqEngine(cfgs, "bf16", "cuda")
    B, T, H, W = args.batch, 4, 100, 154
    ws = eng.acquire(B, T, H, W, True, False)
    eng.pack_weights([torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda") * 0.05 for c in cfgs],
                     [torch.zeros(4 * c.Ch, device="cuda") for c in cfgs])
    eng.forward(ws, torch.randn(B, T, 62, H, W, device="cuda"))
    for l in range(3):
        ws.dG[l].view(torch.bfloat16).normal_(std=0.05)
    g, es = C.byref(ws.g), eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    dh_priv = [torch.zeros_like(ws.dh[l]) for l in range(3)]     # private outputs: the three dgrads must not race

    def fwd(l, st):
        ly = eng.layers[l]
        xs = ws.xs.data_ptr() + B * halo_px * ly.Cxp * es if l == 0 else ws.h[l - 1].data_ptr() + 2 * B * halo_px * ly.Cxp * es
        hs, cs, gs = B * halo_px * ly.Chp * es, B * comp_px * ly.Chp * 4, B * comp_px * 4 * ly.Ch16 * es
        assert lib.nint_cell_fwd(C.byref(ly), g, eng.dt, B, C.c_void_p(xs), C.c_void_p(ws.h[l].data_ptr() + hs),
                                 C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.h[l].data_ptr() + 3 * hs),
                                 C.c_void_p(ws.c[l].data_ptr() + 3 * cs), C.c_void_p(ws.gates[l].data_ptr() + gs),
                                 C.c_void_p(st.cuda_stream)) == 0

    def dgrad(l, st):
        ly = eng.layers[l]
        dgs = B * halo_px * 4 * ly.Ch16 * es
        assert lib.nint_conv_dgrad(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs), None,
                                   C.c_void_p(dh_priv[l].data_ptr()), C.c_void_p(st.cuda_stream)) == 0

    s = [torch.cuda.Stream() for _ in range(3)]
    main_st = torch.cuda.current_stream()

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_st)
        for _ in range(args.iters):
            fn()
        e1.record(main_st); e1.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    def concurrent(kern, order):
        def fn():
            ev = torch.cuda.Event(); ev.record(main_st)
            for l in order:
                s[l].wait_event(ev)
                kern(l, s[l])
            for l in order:
                e = torch.cuda.Event(); e.record(s[l]); main_st.wait_event(e)
        return fn

    for name, kern in (("fwd", fwd), ("dgrad(h cols)", dgrad)):
        seq = timed(lambda: [kern(l, main_st) for l in range(3)])
        each = [timed(lambda l=l: kern(l, main_st)) for l in range(3)]
        c012 = timed(concurrent(kern, (0, 1, 2)))
        c210 = timed(concurrent(kern, (2, 1, 0)))
        print(f"{name}: layers alone {each[0]:.1f} + {each[1]:.1f} + {each[2]:.1f} us; one stream {seq:.1f} us; "
              f"three streams, wide layer first {c012:.1f} us, narrow layers first {c210:.1f} us", flush=True)
        seq12 = timed(lambda: [kern(l, main_st) for l in (1, 2)])
        c12, c21 = timed(concurrent(kern, (1, 2))), timed(concurrent(kern, (2, 1)))
        print(f"{name}: the two narrow layers only: one stream {seq12:.1f} us; two streams {c12:.1f} / {c21:.1f} us", flush=True)


if __name__ == "__main__":
    main()
