"""The two weight-gradient kernel families (csrc/wgrad.hip; autograd of model.py:220) through the C ABI:

    dW[o][c][ky][kx] = sum_{n,y,x} dG[n,y,x,o] * cat[n, y+ky-p, x+kx-p, c],   db[o] = sum dG[., o]

`nint_layer.wide` = 1 forces the 4-wave kernel (64 gate columns per workgroup), 2 the 8-wave kernel (128 gate columns x a
group of channel tiles per workgroup) wherever it is instantiated.  Both reduce the SAME bf16 slabs, so both are compared
with an f64 reference computed from exactly those slab values (torch.nn.grad.conv2d_weight on the unpacked slabs): only
the f32 summation order separates them -- tolerance 2e-5 of the gradient's max (measured: see the printed lines).  Shapes:
ragged grids (rows not a multiple of 4, columns not a multiple of 32), 62 -> 64 padded input channels, h parts that skip
the zero-state time step, several split-K workgroups per column, reductions that leave some splits one tile short.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

# (Cx, Ch, k, N images, H, W, h_skip via has_init=False and B)
SHAPES = {
    "cfg3-like-62-128": (62, 128, 3, 4, 30, 70),
    "64-64-ragged": (64, 64, 3, 6, 21, 45),
    "128-64-tiny-grid": (128, 64, 3, 3, 7, 33),
    "64-128-one-image": (64, 128, 3, 1, 50, 154),
    "refstack-layer1-64-32-x-source-only": (64, 32, 3, 5, 26, 52),
    "bench-layer0-62-64-k5": (62, 64, 5, 3, 28, 77),
    "32-32-k5-ragged": (32, 32, 5, 5, 13, 41),
    "126-64-k5": (126, 64, 5, 2, 22, 64),
    "16-32-k7": (16, 32, 7, 3, 19, 50),
    "62-64-k7": (62, 64, 7, 2, 12, 35),
}


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as p
    return p.load_library()


def _run(lib, eng, ws, l, wide, N, skip):
    from nasa_niswan_amd import _lib
    ly, cfg = eng.layers[l], eng.cfgs[l]
    ly.wide = wide
    dW = torch.full((4 * cfg.Ch, cfg.Cx + cfg.Ch, cfg.k, cfg.k), float("nan"), device="cuda")
    db = torch.full((4 * cfg.Ch,), float("nan"), device="cuda")
    # (a workspace of its own, sized for THIS family: the engine's was sized for the library's choice)
    part = torch.full((lib.nint_wgrad_workspace_bytes(C.byref(ly), eng.dt, eng.n_cu) // 4 + 16,), float("nan"), device="cuda")
    g = ws.g
    halo_px = g.Hh * g.Wh
    es = eng.es
    _lib.check(lib.nint_conv_wgrad(C.byref(ly), C.byref(g), eng.dt, N, C.c_void_p(ws.dG[l].data_ptr()), C.c_void_p(ws.xs.data_ptr()),
                                   C.c_void_p(ws.h[l].data_ptr()), C.c_void_p(dW.data_ptr()), C.c_void_p(db.data_ptr()),
                                   C.c_void_p(part.data_ptr()), part.numel() * 4, eng.n_cu, None), "nint_conv_wgrad")
    torch.cuda.synchronize()
    return dW, db


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_both_weight_gradient_families_reduce_the_same_slabs_to_the_f64_sum(lib, name):
    from nasa_niswan_amd import engine
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    Cx, Ch, k, N, H, W = SHAPES[name]
    torch.manual_seed(5)
    engine.XFOLD = False                    # plain channel-padded x slabs (a thin first-layer input would be fed folded)
    try:
        eng = SeqEngine([LayerCfg(Cx, Ch, k)], "bf16", "cuda")
    finally:
        engine.XFOLD = True
    # one workspace with T = N, B = 1: image n of every slab is "time step n"
    ws = eng.acquire(1, N, H, W, True, True)
    g = ws.g
    P = g.P
    Cxp, Ch16, Chp = eng.cfgs[0].padded(eng.kc)
    bf = torch.bfloat16
    # random slabs, written through views of the interior only (halo / slack / channel padding stay zero)
    xs = ws.xs.view(bf).view(N, g.Hh, g.Wh, Cxp)
    hh = ws.h[0].view(bf).view(N + 1, g.Hh, g.Wh, Chp)
    dG = ws.dG[0].view(bf).view(N, g.Hh, g.Wh, 4 * Ch16)
    xs[:, P:P + H, P:P + W, :Cx] = torch.randn(N, H, W, Cx, device="cuda").to(bf)
    hh[:, P:P + H, P:P + W, :Ch] = (0.5 * torch.randn(N + 1, H, W, Ch, device="cuda")).to(bf)
    dG[:, P:P + H, P:P + W, :] = (0.1 * torch.randn(N, H, W, 4 * Ch16, device="cuda")).to(bf)
    # f64 reference from the slab values: gate-stash column (cblock*4+gate)*16+col -> out channel gate*Ch + cblock*16+col
    dGr = dG[:, P:P + H, P:P + W, :].double().reshape(N, H, W, Ch16 // 16, 4, 16).permute(0, 4, 3, 5, 1, 2).reshape(N, 4 * Ch16, H, W)
    assert Ch16 == Ch
    cat = torch.cat([xs[:, P:P + H, P:P + W, :Cx], hh[:N, P:P + H, P:P + W, :Ch]], dim=3).double().permute(0, 3, 1, 2).contiguous()
    ref_dW = torch.nn.grad.conv2d_weight(cat.cpu(), (4 * Ch, Cx + Ch, k, k), dGr.cpu().contiguous(), padding=k // 2)
    ref_db = dGr.sum(dim=(0, 2, 3)).cpu()
    out = {}
    for wide in (1, 2):
        dW, db = _run(lib, eng, ws, 0, wide, N, 0)
        assert bool(torch.isfinite(dW).all()) and bool(torch.isfinite(db).all()), (name, wide, "unwritten gradient elements")
        eW = float((dW.cpu().double() - ref_dW).abs().max() / ref_dW.abs().max())
        eb = float((db.cpu().double() - ref_db).abs().max() / ref_db.abs().max())
        print(f"  {name} wide={wide}: dW max err / max {eW:.2e}, db {eb:.2e}")
        assert eW <= 2e-5 and eb <= 2e-5, (name, wide, eW, eb)
        out[wide] = (dW, db)
        # bitwise reproducible (fixed fold order, no float atomics)
        dW2, db2 = _run(lib, eng, ws, 0, wide, N, 0)
        assert torch.equal(dW, dW2) and torch.equal(db, db2), (name, wide, "not reproducible")
    eng.layers[0].wide = 0
    eng.release(ws)


def test_the_library_picks_the_128_column_kernel_where_it_is_instantiated(lib):
    """Host arithmetic of the choice, read off the workspace size (the two families size their split-K slabs differently):
    wide = 0 equals wide = 2 where the 8-wave kernel is instantiated and wide = 1 everywhere else."""
    from nasa_niswan_amd._lib import NintLayer, NINT_BF16, NINT_F32

    def ws_bytes(Cx, Ch, k, wide, dt=NINT_BF16, xfold=0):
        ly = NintLayer()
        kc = 32 if dt == NINT_BF16 else 16
        rup = lambda a, b: (a + b - 1) // b * b
        ly.Cx, ly.Ch, ly.k, ly.xfold, ly.wide = Cx, Ch, k, xfold, wide
        ly.Cxp, ly.Ch16, ly.Chp = rup(k * Cx if xfold else Cx, kc), rup(Ch, 16), rup(Ch, kc)
        return lib.nint_wgrad_workspace_bytes(C.byref(ly), dt, 256)

    for Cx, Ch, k, held in ((62, 128, 3, True), (128, 128, 3, True), (64, 64, 3, True), (64, 32, 3, True), (32, 32, 3, False), (32, 16, 3, False),
                            (62, 64, 5, True), (126, 64, 5, True), (64, 32, 5, True), (64, 16, 5, False), (62, 64, 7, True),
                            (62, 64, 1, False), (62, 48, 3, False)):
        b0, b1, b2 = (ws_bytes(Cx, Ch, k, w) for w in (0, 1, 2))
        assert b0 > 0 and b1 > 0 and b2 > 0
        assert (b0 == b2) and ((b2 != b1) == held), (Cx, Ch, k, b0, b1, b2)
    assert ws_bytes(62, 128, 3, 2, NINT_F32) == ws_bytes(62, 128, 3, 1, NINT_F32)      # f32 storage: the 4-wave kernel only
