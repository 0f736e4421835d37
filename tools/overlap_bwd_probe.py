#!/usr/bin/env python3
"""Which launches of one BPTT step would gain from running side by side?  Two HIP streams, bench shape.
Pairs: pointwise backward of layer 0 with the top layer's fused step (both HBM / latency bound), layer-1 dgrad with layer-0 dgrad
(both on the matrix pipe), layer-1 dgrad with layer-0 pointwise.  Private outputs, so that nothing races.
    python tools/overlap_bwd_probe.py [--batch 8] [--iters 30]"""
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
        ws.dh[l].view(torch.bfloat16).normal_(std=0.05); ws.dc[l].normal_(std=0.05)
    g, es = C.byref(ws.g), eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    dh_priv = [torch.zeros_like(ws.dh[l]) for l in range(3)]
    dx_priv = [torch.zeros_like(ws.dh[max(l - 1, 0)]) for l in range(3)]
    dc_priv = [ws.dc[l].clone() for l in range(3)]
    dG_priv = [torch.zeros(B * halo_px * 4 * eng.layers[l].Ch16 * es, dtype=torch.uint8, device="cuda") for l in range(3)]

    def pw(l, st):
        ly = eng.layers[l]
        cs, gs = B * comp_px * ly.Chp * 4, B * comp_px * 4 * ly.Ch16 * es
        assert lib.nint_cell_bwd_pointwise(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.gates[l].data_ptr() + gs),
                                           C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.c[l].data_ptr() + 2 * cs),
                                           C.c_void_p(ws.dh[l].data_ptr()), C.c_void_p(dc_priv[l].data_ptr()),
                                           C.c_void_p(dG_priv[l].data_ptr()), C.c_void_p(st.cuda_stream)) == 0

    def dgrad(l, st):
        ly = eng.layers[l]
        dgs = B * halo_px * 4 * ly.Ch16 * es
        assert lib.nint_conv_dgrad(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs),
                                   C.c_void_p(dx_priv[l].data_ptr()) if l > 0 else None,
                                   C.c_void_p(dh_priv[l].data_ptr()), C.c_void_p(st.cuda_stream)) == 0

    def fused(l, st):
        ly = eng.layers[l]
        dgs = B * halo_px * 4 * ly.Ch16 * es
        cs, gs = B * comp_px * ly.Chp * 4, B * comp_px * 4 * ly.Ch16 * es
        assert lib.nint_cell_bwd_fused(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs),
                                       C.c_void_p(dx_priv[l].data_ptr()), C.c_void_p(ws.gates[l].data_ptr() + gs),
                                       C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.c[l].data_ptr() + 2 * cs),
                                       C.c_void_p(dh_priv[l].data_ptr()) if l < 2 else None, C.c_void_p(dc_priv[l].data_ptr()),
                                       C.c_void_p(dG_priv[l].data_ptr()), C.c_void_p(st.cuda_stream)) == 0

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

    def concurrent(jobs):
        def fn():
            ev = torch.cuda.Event(); ev.record(main_st)
            for i, (k, l) in enumerate(jobs):
                s[i].wait_event(ev)
                k(l, s[i])
            for i in range(len(jobs)):
                e = torch.cuda.Event(); e.record(s[i]); main_st.wait_event(e)
        return fn

    def serial_marked(jobs):       # the same event traffic, one job after the other: what the markers cost by themselves
        def fn():
            for i, (k, l) in enumerate(jobs):
                ev = torch.cuda.Event(); ev.record(main_st)
                s[i].wait_event(ev)
                k(l, s[i])
                e = torch.cuda.Event(); e.record(s[i]); main_st.wait_event(e)
        return fn

    names = {pw: "pointwise", dgrad: "dgrad", fused: "fused"}
    pairs = [((pw, 0), (fused, 2)), ((pw, 0), (fused, 2), (pw, 1)), ((pw, 0), (pw, 1)), ((dgrad, 0), (dgrad, 1)),
             ((dgrad, 1), (pw, 0)), ((dgrad, 0), (fused, 2)), ((dgrad, 0), (pw, 1))]
    for jobs in pairs:
        each = [timed(lambda k=k, l=l: k(l, main_st)) for k, l in jobs]
        seq = timed(lambda: [k(l, main_st) for k, l in jobs])
        con = timed(concurrent(jobs))
        rev = timed(concurrent(jobs[::-1]))
        ser = timed(serial_marked(jobs))
        label = " | ".join(f"{names[k]}{l}" for k, l in jobs)
        print(f"{label}: alone {' + '.join(f'{e:.1f}' for e in each)} us; one stream {seq:.1f}; streams {con:.1f} / reversed issue {rev:.1f}; "
              f"serial with the same markers {ser:.1f}", flush=True)


if __name__ == "__main__":
    main()
