"""GPU: the fused BPTT step (nint_cell_bwd_fused: conv backward-data of time t+1 with the pointwise LSTM backward of
time t in its epilogue) against the two separate launches it replaces (nint_conv_dgrad -> dh, nint_cell_bwd_pointwise),
on the bench stack's three layers (8-row / 4-row tiles, 1-4 K-slices, x columns present and absent, ragged grids).
f32: same arithmetic, only the association of dh = (h columns) + (x columns of the layer above) is shared -> 1e-6;
bf16: the fused step keeps dh in f32 where the pair rounds it to bf16 in between -> rel-L2 1e-2 (the pair is the one
further from the oracle)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as p
    p.load_library()
    return p


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,B", [(100, 154, 2), (11, 19, 3), (8, 16, 1)])
def test_fused_step_equals_dgrad_then_pointwise(pkg, dtype, H, W, B):
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    lib = pkg.load_library()
    torch.manual_seed(0)
    hidden, ks, cin = (64, 32, 16), (5, 3, 3), 62
    cfgs = []
    for ch, k in zip(hidden, ks):
        cfgs.append(LayerCfg(cin, ch, k)); cin = ch
    eng = SeqEngine(cfgs, dtype, "cuda")
    T = 2
    ws = eng.acquire(B, T, H, W, True, False)
    eng.pack_weights([torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda") * 0.05 for c in cfgs],
                     [torch.randn(4 * c.Ch, device="cuda") * 0.1 for c in cfgs])
    eng.forward(ws, torch.randn(B, T, 62, H, W, device="cuda"))
    et = torch.bfloat16 if eng.es == 2 else torch.float32
    g, es = C.byref(ws.g), eng.es
    halo_px, comp_px = ws.g.Hh * ws.g.Wh, H * W
    P = ws.g.P
    for l, cfg in enumerate(cfgs):
        ly = eng.layers[l]
        Gc = 4 * ly.Ch16
        # dG of time 1: random in the interior, zero halo (as the pointwise backward leaves it)
        dG = ws.dG[l].view(et).view(T * B, ws.g.Hh, ws.g.Wh, Gc)
        dG.zero_()
        dG[B:, P:P + H, P:P + W] = (torch.randn(B, H, W, Gc, device="cuda") * 0.1).to(et)
        dgs, gs, cs = B * halo_px * Gc * es, B * comp_px * Gc * es, B * comp_px * ly.Chp * 4
        above = (torch.randn(B * comp_px * ly.Chp, device="cuda") * 0.1).to(et) if l < 2 else None
        dc0 = torch.randn(B * comp_px * ly.Chp, device="cuda") * 0.1
        vp = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
        # --- the pair
        dh = torch.zeros(B * comp_px * ly.Chp, device="cuda", dtype=et)
        dx_ref = torch.zeros(B * comp_px * ly.Cxp, device="cuda", dtype=et) if l > 0 else None
        assert lib.nint_conv_dgrad(C.byref(ly), g, eng.dt, B, vp(ws.dG[l], dgs), vp(dx_ref) if l > 0 else None, vp(dh), None) == 0
        if above is not None:
            dh = (dh.float() + above.float()).to(et)
        dc_ref = dc0.clone()
        dG_ref = torch.zeros(B * halo_px * Gc, device="cuda", dtype=et)
        assert lib.nint_cell_bwd_pointwise(C.byref(ly), g, eng.dt, B, vp(ws.gates[l]), vp(ws.c[l]), vp(ws.c[l], cs),
                                           vp(dh), vp(dc_ref), vp(dG_ref), None) == 0
        # --- fused
        dc = dc0.clone()
        dx = torch.full((B * comp_px * ly.Cxp,), 7.0, device="cuda", dtype=et) if l > 0 else None      # stored, not accumulated
        assert lib.nint_cell_bwd_fused(C.byref(ly), g, eng.dt, B, vp(ws.dG[l], dgs), vp(dx) if l > 0 else None, vp(ws.gates[l]),
                                       vp(ws.c[l]), vp(ws.c[l], cs), vp(above) if above is not None else None, vp(dc),
                                       vp(ws.dG[l]), None) == 0
        torch.cuda.synchronize()
        dG_f = ws.dG[l].view(et)[:B * halo_px * Gc]
        pairs = {"dG": (dG_f, dG_ref), "dc": (dc, dc_ref)}
        if l > 0:
            pairs["dx"] = (dx, dx_ref)
        for name, (a, b) in pairs.items():
            a, b = a.double(), b.double()
            assert torch.isfinite(a).all(), (l, name)
            if dtype == "f32":
                err, ref = float((a - b).abs().max()), float(b.abs().max())
                assert err <= 1e-6 * ref + 1e-9, (l, name, err, ref)
            elif name == "dx":
                # the same f32 sums, rounded to bf16 once -- but small batches split the classic launch's columns over more
                # workgroups (another K-slice order), so a sum may land on the other side of a bf16 rounding boundary
                r = float((a - b).norm() / b.norm())
                assert r <= 1e-3 and float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()), (l, name, r)
            else:
                r = float((a - b).norm() / b.norm())
                assert r <= 1e-2, (l, name, r)
        # the halo ring of the dG slab stays zero
        d4 = dG_f.view(B, ws.g.Hh, ws.g.Wh, Gc).float()
        assert float(d4[:, :P].abs().max()) == 0 and float(d4[:, :, :P].abs().max()) == 0
        assert float(d4[:, P + H:].abs().max()) == 0 and float(d4[:, :, P + W:].abs().max()) == 0


@pytest.mark.parametrize("mode", [1, 2, 0x40000404, 0x40000202, 0x40000005, 0x40020404, 0x40060000])
def test_backward_schedules_keep_parity(pkg, mode):
    """nint_seq.fuse_bwd: 1 = every layer runs the classic pair, 2 = every layer runs the fused step, one time step behind
    the layer above (the default, 0, mixes them per layer and is what the rest of the suite runs); 0x40000000 | masks =
    explicit: ..404 layer 2 fused and running layer 1's pointwise backward on its x columns, ..202 the same one layer
    down, ..005 layers 0 and 2 fused, layer 1 classic, 0x4002.... / 0x4006....: classic layers 1 (and 2) run the pointwise
    backward of the classic layer below on their x columns.  Every gradient, the input gradient included, against the oracle on
    ragged / multi-layer / reference-size shapes, T = 1 .. 3."""
    from nasa_niswan_amd import engine
    from test_gpu_shapes import CASES, check, run_case
    engine.FUSE_BWD = mode
    try:
        for name in ("ragged-grid-odd-channels", "batch1-T1", "k1-and-k3", "wide-hidden-48"):
            for dtype in ("f32", "bf16"):
                check(run_case(pkg, *CASES[name], dtype), dtype)
        check(run_case(pkg, 5, [64, 32, 16], [5, 3, 3], 1, 2, 3, 20, 36, "f32"), "f32")        # the reference stack, T = 3
        check(run_case(pkg, 62, [64, 32, 16], [5, 3, 3], 20, 1, 2, 100, 154, "bf16"), "bf16")  # bench geometry, T = 2
        check(run_case(pkg, 6, [128, 128], [3, 3], 1, 1, 2, 10, 18, "bf16"), "bf16")           # several column groups per launch
    finally:
        engine.FUSE_BWD = 0


def test_fused_and_classic_schedules_agree(pkg):
    """Same model and inputs through both schedules, f32: every gradient (input gradient included) agrees to rounding."""
    from nasa_niswan_amd import engine
    torch.manual_seed(3)
    C_, hidden, ks, B, T, H, W = 6, [16, 8], [3, 3], 2, 3, 12, 20
    X = torch.randn(B, T, C_, H, W, device="cuda")
    wgt = torch.randn(B, 1, H, W, device="cuda")
    res = {}
    for mode in (1, 2):
        engine.FUSE_BWD = mode
        try:
            torch.manual_seed(4)
            net = pkg.ConvLSTM(C_, hidden, ks, 2).cuda()
            Xg = X.clone().requires_grad_(True)
            (net(Xg) * wgt).sum().backward()
            res[mode] = [Xg.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        finally:
            engine.FUSE_BWD = 0
    for a, b in zip(res[1], res[2]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9


@pytest.mark.parametrize("wave", [None, 4, 0])
@pytest.mark.parametrize("mode", [1, 2, 0x40000404, 0x40000003])
def test_schedules_with_a_given_initial_state_multi_layer(pkg, mode, wave):
    """nint_seq.has_init_state with fused layers BELOW the top (the combination the model suite only reaches with one
    layer): at u == 0 a fused layer runs the plain dgrad and stores d/dh_init into a dh buffer that, for l < L-1, also
    carried the upper layer's x columns.  Three layers, T = 3, given h0 / c0 per layer; every weight / bias gradient, the input
    gradient and d/dh_init, d/dc_init of every layer against the oracle's autograd (model.py:253-271 from a given state).
    Under the engine's launch schedule for this size (nint_seq.wave = 5: with classic steps in layers 0 and 1 the bottom layer's
    d/dh travels in two pieces and its d/dh_init must still land in dh[0]), under wave = 4 and in the time-major order."""
    from nasa_niswan_amd import engine
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    from oracle import convlstm_oracle as O
    Cin, hidden, ks, B, T, H, W = 5, [16, 8, 8], [3, 3, 3], 2, 3, 11, 19
    L = 3
    params = O.synth_params(Cin, hidden, ks, L, seed=9)
    rng = np.random.default_rng(9)
    X = torch.from_numpy(rng.standard_normal((B, T, Cin, H, W)).astype(np.float32))
    h0 = [torch.from_numpy(rng.standard_normal((B, c, H, W)).astype(np.float32)) * 0.5 for c in hidden]
    c0 = [torch.from_numpy(rng.standard_normal((B, c, H, W)).astype(np.float32)) * 0.5 for c in hidden]
    wgt = torch.from_numpy(rng.standard_normal((B, 1, H, W)).astype(np.float32))
    # oracle
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Xo = X.clone().requires_grad_(True)
    h0o = [t.clone().requires_grad_(True) for t in h0]
    c0o = [t.clone().requires_grad_(True) for t in c0]
    po = O.convlstm_forward(Xo, leaf, h0=h0o, c0=c0o)
    (po * wgt).sum().backward()
    engine.FUSE_BWD, engine.FORCE_WAVE = mode, wave
    try:
        for dtype in ("f32", "bf16"):
            eng = SeqEngine([LayerCfg(Cin if l == 0 else hidden[l - 1], hidden[l], ks[l]) for l in range(L)], dtype, "cuda")
            ws = eng.acquire(B, T, H, W, True, True)
            assert ws.seq.has_init_state == 1 and ws.seq.fuse_bwd == mode
            eng.pack_weights([params[f"layers.{l}.conv.weight"].cuda() for l in range(L)],
                             [params[f"layers.{l}.conv.bias"].cuda() for l in range(L)])
            eng.forward(ws, X.cuda(), [t.cuda() for t in h0], [t.cuda() for t in c0])
            w_head, b_head = params["conv.weight"].cuda(), params["conv.bias"].cuda()
            pred = eng.head_forward(ws, w_head, b_head)
            eng.head_backward(ws, w_head, wgt.cuda().expand(B, 1, H, W).contiguous())
            for l in range(L):
                if l < L - 1:
                    eng.set_state_grads(ws, l, None, None)
                else:
                    ws.dc[l].zero_()
            dWs, dbs, dx = eng.backward(ws, True)
            res = {"pred": (pred.cpu(), po.detach()), "dX": (dx.cpu(), Xo.grad)}
            for l in range(L):
                res[f"dW{l}"] = (dWs[l].cpu(), leaf[f"layers.{l}.conv.weight"].grad)
                res[f"db{l}"] = (dbs[l].cpu(), leaf[f"layers.{l}.conv.bias"].grad)
                dh, dc = eng.state_grads(ws, l)
                res[f"dh0_{l}"] = (dh.cpu(), h0o[l].grad)
                res[f"dc0_{l}"] = (dc.cpu(), c0o[l].grad)
            eng.release(ws)
            for k, (a, b) in res.items():
                a, b = a.double(), b.double()
                assert torch.isfinite(a).all(), (dtype, k)
                if dtype == "f32":
                    err, ref = float((a - b).abs().max()), float(b.abs().max())
                    assert err <= 1e-3 * ref + 1e-5, (dtype, k, err, ref)
                else:
                    r = float((a - b).norm() / (b.norm() + 1e-30))
                    assert r <= 2e-2, (dtype, k, r)
    finally:
        engine.FUSE_BWD, engine.FORCE_WAVE = 0, None
