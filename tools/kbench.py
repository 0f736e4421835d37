#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the bench workload's shapes (HIP events on the launch stream).
    python tools/kbench.py [--dtype bf16] [--batch 8] [--iters 20] [--only fwd0,dgrad0,...]
Prints one line per kernel: average ms, algorithmic TFLOP/s (or GB/s)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nasa_niswan_amd as pkg  # noqa: E402
from nasa_niswan_amd.engine import LayerCfg, SeqEngine  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--T", type=int, default=12)
    ap.add_argument("--C", type=int, default=62)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--H", type=int, default=100)
    ap.add_argument("--W", type=int, default=154)
    ap.add_argument("--zeros", action="store_true", help="all-zero inputs and weights (data-dependent power / clock check)")
    ap.add_argument("--lib", default=None, help="load this build of the library instead of the product one (A/B copies, stamp builds)")
    ap.add_argument("--wide", type=int, default=-1, help="nint_layer.wide of every layer: weight-gradient kernel family (-1: the engine's choice)")
    ap.add_argument("--tile-rows", type=int, default=0, help="force 4- or 8-row tiles for the gate / dgrad kernels")
    ap.add_argument("--hidden", default="64,32,16", help="hidden widths of the three layers (BASELINE configs[3]: 128,128,128)")
    ap.add_argument("--ks", default="5,3,3", help="kernel sizes of the three layers (configs[3]: 3,3,3)")
    ap.add_argument("--split", type=int, default=1, help="issue the forward gate kernel as this many launches over image groups")
    args = ap.parse_args()
    lib = pkg.load_library(args.lib) if args.lib else pkg.load_library()
    hidden, ks = tuple(int(v) for v in args.hidden.split(",")), tuple(int(v) for v in args.ks.split(","))
    cfgs, cin = [], args.C
    for ch, k in zip(hidden, ks):
        cfgs.append(LayerCfg(cin, ch, k)); cin = ch
    from nasa_niswan_amd import engine as _engine
    _engine.FORCE_TILE_ROWS = args.tile_rows
    eng = SeqEngine(cfgs, args.dtype, "cuda")
    for ly in eng.layers:
        ly.tile_rows = args.tile_rows
        if args.wide >= 0:
            ly.wide = args.wide
    B, T, H, W = args.batch, args.T, args.H, args.W
    ws = eng.acquire(B, T, H, W, True, False)
    ws_w = [torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda") * (0.0 if args.zeros else 0.05) for c in cfgs]
    ws_b = [torch.zeros(4 * c.Ch, device="cuda") for c in cfgs]
    eng.pack_weights(ws_w, ws_b)
    X = torch.randn(B, T, args.C, H, W, device="cuda") * (0.0 if args.zeros else 1.0)
    eng.forward(ws, X)       # fills every slab with realistic (random-data) values
    for l in range(len(cfgs)):
        ws.dh[l].view(torch.bfloat16 if eng.es == 2 else torch.float32).normal_(); ws.dc[l].normal_()
    g, st = C.byref(ws.g), None
    es = eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    only = set(args.only.split(",")) if args.only else None
    rows = []

    def run(name, fn, flops=None, bytes_=None):
        if only and name not in only:
            return
        ms = timeit(fn, args.iters)
        extra = f"{flops / ms / 1e9:8.1f} TFLOP/s" if flops else (f"{bytes_ / ms / 1e6:8.1f} GB/s" if bytes_ else "")
        print(f"{name:12s} {ms * 1e3:9.1f} us  {extra}", flush=True)
        rows.append((name, ms))

    for l, cfg in enumerate(cfgs):
        ly = eng.layers[l]
        xs = ws.xs.data_ptr() + 1 * B * halo_px * ly.Cxp * es if l == 0 else ws.h[l - 1].data_ptr() + 2 * B * halo_px * ly.Cxp * es
        hs = B * halo_px * ly.Chp * es
        cs = B * comp_px * ly.Chp * 4
        gs = B * comp_px * 4 * ly.Ch16 * es
        dgs = B * halo_px * 4 * ly.Ch16 * es
        fl = 2.0 * B * H * W * cfg.k ** 2 * (cfg.Cx + cfg.Ch) * 4 * cfg.Ch

        def fwd(ly=ly, xs=xs, hs=hs, cs=cs, gs=gs, l=l):
            nb = B // args.split
            for part in range(args.split):      # image groups: every slab pointer advances by the group's images
                f = part * nb
                assert lib.nint_cell_fwd(C.byref(ly), g, eng.dt, nb if part < args.split - 1 else B - f,
                                         C.c_void_p(xs + f * halo_px * ly.Cxp * es),
                                         C.c_void_p(ws.h[l].data_ptr() + hs + f * halo_px * ly.Chp * es),
                                         C.c_void_p(ws.c[l].data_ptr() + cs + f * comp_px * ly.Chp * 4),
                                         C.c_void_p(ws.h[l].data_ptr() + 2 * hs + f * halo_px * ly.Chp * es),
                                         C.c_void_p(ws.c[l].data_ptr() + 2 * cs + f * comp_px * ly.Chp * 4),
                                         C.c_void_p(ws.gates[l].data_ptr() + gs + f * comp_px * 4 * ly.Ch16 * es), st) == 0
        run(f"fwd{l}", fwd, fl)

        def pw(ly=ly, cs=cs, gs=gs, dgs=dgs, l=l):
            assert lib.nint_cell_bwd_pointwise(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.gates[l].data_ptr() + gs),
                                               C.c_void_p(ws.c[l].data_ptr() + cs), C.c_void_p(ws.c[l].data_ptr() + 2 * cs),
                                               C.c_void_p(ws.dh[l].data_ptr()), C.c_void_p(ws.dc[l].data_ptr()),
                                               C.c_void_p(ws.dG[l].data_ptr() + dgs), st) == 0
        run(f"pointwise{l}", pw, None, B * comp_px * ly.Ch16 * (9 * es + 16))   # gates+dG (ET), dh (ET), c_prev, c_new, dc r+w (f32)
    # fill dG for every t so that wgrad sees random data
    for l in range(len(cfgs)):
        ws.dG[l].view(torch.bfloat16 if es == 2 else torch.float32).normal_(std=0.05)
    for l, cfg in enumerate(cfgs):
        ly = eng.layers[l]
        dgs = B * halo_px * 4 * ly.Ch16 * es
        dx = ws.dh[l - 1].data_ptr() if l > 0 else None
        ncols = cfg.Ch + (cfg.Cx if l > 0 else 0)
        fl = 2.0 * B * H * W * cfg.k ** 2 * 4 * cfg.Ch * ncols

        def dg(ly=ly, dgs=dgs, dx=dx, l=l):
            assert lib.nint_conv_dgrad(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs),
                                       C.c_void_p(dx) if dx else None, C.c_void_p(ws.dh[l].data_ptr()), st) == 0
        run(f"dgrad{l}", dg, fl)
        gs, cs = B * comp_px * 4 * ly.Ch16 * es, B * comp_px * ly.Chp * 4

        def fused(ly=ly, dgs=dgs, dx=dx, l=l, gs=gs, cs=cs):      # dgrad(t+1) + pointwise backward(t) in one launch
            assert lib.nint_cell_bwd_fused(C.byref(ly), g, eng.dt, B, C.c_void_p(ws.dG[l].data_ptr() + dgs), C.c_void_p(dx) if dx else None,
                                           C.c_void_p(ws.gates[l].data_ptr() + gs), C.c_void_p(ws.c[l].data_ptr() + cs),
                                           C.c_void_p(ws.c[l].data_ptr() + 2 * cs), C.c_void_p(ws.dh[l].data_ptr()) if l < len(cfgs) - 1 else None,
                                           C.c_void_p(ws.dc[l].data_ptr()), C.c_void_p(ws.dG[l].data_ptr()), st) == 0
        run(f"fused{l}", fused, fl)
        dW = torch.empty(4 * cfg.Ch, cfg.Cx + cfg.Ch, cfg.k, cfg.k, device="cuda")
        db = torch.empty(4 * cfg.Ch, device="cuda")
        x_all = ws.xs.data_ptr() if l == 0 else ws.h[l - 1].data_ptr() + B * halo_px * ly.Cxp * es
        flw = 2.0 * T * B * H * W * cfg.k ** 2 * (cfg.Cx + cfg.Ch) * 4 * cfg.Ch

        def wg(ly=ly, x_all=x_all, dW=dW, db=db, l=l):
            assert lib.nint_conv_wgrad(C.byref(ly), g, eng.dt, T * B, C.c_void_p(ws.dG[l].data_ptr()), C.c_void_p(x_all),
                                       C.c_void_p(ws.h[l].data_ptr()), C.c_void_p(dW.data_ptr()), C.c_void_p(db.data_ptr()),
                                       C.c_void_p(eng.wg_partial.data_ptr()), eng.wg_partial.numel() * 4, eng.n_cu, st) == 0
        run(f"wgrad{l}", wg, flw)
    xs_pack = lambda: lib.nint_pack_btchw(C.c_void_p(X.data_ptr()), C.c_void_p(ws.xs.data_ptr()), B, T, args.C, ws.Cxp0, g, eng.dt, st)
    run("pack", xs_pack, None, X.numel() * 4 + B * T * comp_px * ws.Cxp0 * es)
    if args.C % 3 == 2 and (not only or only & {"preproc_slab", "preproc_nchw"}):
        # a-6: fuse + z-score + halo pad of a whole batch in one launch, from a record resident in HBM
        from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
        ds = SyntheticE33OMA_CRNN("train", padding=(H, W), in_channels=args.C, sequence_length=T, levels=(args.C - 2) // 3,
                                  n_steps=T + 2 * B + 4, grid=(H - 10, W - 10), device="cuda")
        idx = list(range(0, 2 * B, 2))
        sb, _ = ds.slab_batch(idx)
        alg = B * T * args.C * H * W
        run("preproc_slab", lambda: sb.fill_slab(eng, ws), None, alg * (4 + es))     # 4 B in + ET out per element
        run("preproc_nchw", lambda: ds.device_batch(idx), None, alg * 8)
    tot = sum(ms * (T if not n.startswith(("wgrad", "pack", "preproc")) else 1) for n, ms in rows if n != "preproc_nchw")
    print(f"sum over a step (T x per-step kernels + wgrad + pack): {tot:.2f} ms")


if __name__ == "__main__":
    main()
