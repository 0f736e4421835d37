"""GPU parity on ragged / padded / wide shapes (every padding rule of the slab layout is exercised):
grids that are not multiples of the 8x16 pixel tile, channel counts that are not multiples of
16/32, hidden widths that need several column groups, wide heads, B=1, T=1 -- plus the reduced-size
versions of BASELINE.json configs[3] (3 x hidden 128) and configs[4] (126 inputs, 200 outputs), and
size-independent properties at the full bench size."""
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


def run_case(pkg, C, hidden, ks, out, B, T, H, W, dtype, seed=0):
    from oracle import convlstm_oracle as O
    params = O.synth_params(C, hidden, ks, len(hidden), out_channels=out, seed=seed)
    rng = np.random.default_rng(seed + 5)
    X = torch.from_numpy(rng.standard_normal((B, T, C, H, W)).astype(np.float32))
    wgt = torch.from_numpy(rng.standard_normal((B, out, H, W)).astype(np.float32))
    net = pkg.ConvLSTM(C, hidden, ks, len(hidden), out_channels=out, compute_dtype=dtype).cuda()
    net.load_state_dict(params)
    Xd = X.cuda().requires_grad_(True)
    pred = net(Xd)
    (pred * wgt.cuda()).sum().backward()
    torch.cuda.synchronize()
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Xo = X.clone().requires_grad_(True)
    po = O.convlstm_forward(Xo, leaf)
    (po * wgt).sum().backward()
    res = {"pred": (pred.detach().cpu(), po.detach()), "dX": (Xd.grad.cpu(), Xo.grad)}
    for k, p in net.named_parameters():
        res["grad." + k] = (p.grad.cpu(), leaf[k].grad)
    return res


def check(res, dtype):
    for k, (a, b) in res.items():
        a, b = a.double().numpy(), b.double().numpy()
        if dtype == "f32":
            err, ref = np.abs(a - b).max(), np.abs(b).max()
            print(f"  {k}: max abs err {err:.2e} (ref max {ref:.2e})")
            assert err <= 1e-3 * ref + 1e-5, (k, err, ref)
        else:
            r = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
            print(f"  {k}: rel-L2 {r:.2e}")
            assert r <= 2e-2, (k, r)


CASES = {
    # name: (C, hidden, ks, out, B, T, H, W)
    "ragged-grid-odd-channels": (7, [24, 16], [3, 5], 1, 3, 2, 11, 19),
    "batch1-T1": (5, [16], [5], 1, 1, 1, 9, 17),
    "k1-and-k3": (4, [16, 8], [1, 3], 2, 2, 2, 8, 16),
    "cfg3-reduced (3 x hidden 128, k3)": (6, [128, 128, 128], [3, 3, 3], 1, 1, 2, 10, 18),
    "cfg4-reduced (126 in, 200 out)": (126, [64, 32, 16], [5, 3, 3], 200, 1, 2, 12, 20),
    "wide-hidden-48": (3, [48], [3], 3, 2, 2, 13, 33),
    # thick inputs that still fold (ceil(3*65/32) = 7 < 9 K-steps): the pack kernel's channel-row tile is 78 KiB (> 64 KiB:
    # opt-in LDS size) on a 1-degree-wide grid, and 180 KiB (> the CU's LDS: per-element pack path) for 100 channels x 450
    "folded-65-channels-wide-grid": (65, [16], [3], 1, 1, 2, 9, 298),
    "folded-100-channels-450-wide": (100, [16], [3], 1, 1, 1, 8, 450),
    # ONE input channel, folded (channel = kx): the fold's co / C by multiply-high has no 32-bit magic number for C = 1
    # (found by tools/fuzz_shapes.py: the pack kernel read rows past its tile)
    "single-input-channel-k5": (1, [32, 16], [5, 3], 1, 2, 2, 8, 49),
    "single-input-channel-k3-one-column-vectors": (1, [8], [3], 2, 1, 2, 6, 4),
}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_shapes_vs_oracle(pkg, name, dtype):
    check(run_case(pkg, *CASES[name], dtype), dtype)


def test_full_size_properties(pkg):
    """100x154 padded grid, T=12, C=62, out=20 (the bench workload) at B=2: size-independent properties
    on top of the oracle comparison at this size (tests/test_gpu_fullsize.py): run-to-run bitwise
    determinism, batch independence (a sample's result does not depend on its batch mates: bit for bit while the launch
    shape is pinned -- small batches otherwise take smaller tiles / split gate columns, i.e. another f32 summation order,
    and agree to rounding) and agreement of the f32 and bf16 paths."""
    from nasa_niswan_amd import engine
    torch.manual_seed(0)
    C, hidden, ks, out, B, T, H, W = 62, [64, 32, 16], [5, 3, 3], 20, 2, 12, 100, 154
    X = torch.randn(B, T, C, H, W, device="cuda")
    wgt = torch.randn(B, out, H, W, device="cuda")
    nets = {}
    for dt in ("f32", "bf16"):
        torch.manual_seed(1)
        nets[dt] = pkg.ConvLSTM(C, hidden, ks, 3, out_channels=out, compute_dtype=dt).cuda()
    outs, grads = {}, {}
    for dt, net in nets.items():
        for rep in range(2):
            net.zero_grad()
            pred = net(X)
            (pred * wgt).sum().backward()
            g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
            if rep == 0:
                outs[dt], grads[dt] = pred.detach().clone(), g.clone()
            else:
                assert torch.equal(pred, outs[dt]) and torch.equal(g, grads[dt]), f"{dt}: not deterministic"
        with torch.no_grad():
            single = net(X[1:2])
        rel = float((single - outs[dt][1:2]).norm() / outs[dt][1:2].norm())
        assert rel <= (1e-6 if dt == "f32" else 5e-3), f"{dt}: batch dependence {rel}"
    # ... and bit for bit with the tile height pinned (engines are built per module: fresh modules, same weights)
    engine.FORCE_TILE_ROWS = 8
    try:
        for dt in ("f32", "bf16"):
            torch.manual_seed(1)
            net = pkg.ConvLSTM(C, hidden, ks, 3, out_channels=out, compute_dtype=dt).cuda()
            with torch.no_grad():
                both, single = net(X), net(X[1:2])
            assert torch.equal(single, both[1:2]), f"{dt}: batch dependence with the launch shape pinned"
    finally:
        engine.FORCE_TILE_ROWS = 0
    r = float((outs["bf16"] - outs["f32"]).norm() / outs["f32"].norm())
    rg = float((grads["bf16"] - grads["f32"]).norm() / grads["f32"].norm())
    print(f"  bf16 vs f32 at full size: pred rel-L2 {r:.2e}, grads rel-L2 {rg:.2e}")
    assert r < 2e-2 and rg < 2e-2
    assert torch.isfinite(grads["f32"]).all()


def heavy_tail_batch(B, T, H, W, seed=0):
    """SURVEY.md section 8d stress inputs for the ref-pinned 5-channel stack [u, v, omega, prec, src]:
    z-scored precipitation and emission are max(0, lognormal) fields scaled to the extremes of
    variable_statistics.json (max z about 65 and 124); the target reaches z about 165."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, T, 5, H, W)).astype(np.float32)
    for ch, zmax in ((3, 65.0), (4, 124.0)):
        f = rng.lognormal(mean=0.0, sigma=1.5, size=(B, T, H, W))
        X[:, :, ch] = (f / f.max() * zmax).astype(np.float32)
    y = rng.lognormal(mean=0.0, sigma=1.5, size=(B, 1, H, W))
    y = (y / y.max() * 165.0).astype(np.float32)
    return torch.from_numpy(X), torch.from_numpy(y)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_heavy_tail_inputs(pkg, dtype):
    """Forward, MSE+L1 loss and every gradient on heavy-tailed inputs (what bf16 storage has to survive)."""
    from oracle import convlstm_oracle as O
    C, hidden, ks, B, T, H, W = 5, [64, 32, 16], [5, 3, 3], 2, 3, 20, 36
    params = O.synth_params(C, hidden, ks, 3, out_channels=1, seed=3)
    X, y = heavy_tail_batch(B, T, H, W)
    assert X[:, :, 4].max() > 100 and y.max() > 150
    net = pkg.ConvLSTM(C, hidden, ks, 3, compute_dtype=dtype).cuda()
    net.load_state_dict(params)
    pred = net(X.cuda())
    loss = torch.nn.functional.mse_loss(pred, y.cuda()) + torch.nn.functional.l1_loss(pred, y.cuda())
    loss.backward()
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    po = O.convlstm_forward(X, leaf)
    lo = torch.nn.functional.mse_loss(po, y) + torch.nn.functional.l1_loss(po, y)
    lo.backward()
    assert torch.isfinite(pred).all()
    res = {"pred": (pred.detach().cpu(), po.detach()), "loss": (loss.detach().cpu().reshape(1), lo.detach().reshape(1))}
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        res["grad." + k] = (p.grad.cpu(), leaf[k].grad)
    check(res, dtype)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_k7_trains(pkg, dtype):
    """The reference accepts any odd kernel size (padding k//2, model.py:204).  7x7: the weight gradient's 49 taps run as
    two column groups; a thin 3-channel input is folded (7 x 21 channels), a 16-channel one is not."""
    check(run_case(pkg, 3, [8, 8], [7, 7], 1, 2, 2, 12, 20, dtype, seed=7), dtype)
    check(run_case(pkg, 16, [16], [7], 2, 1, 2, 9, 37, dtype, seed=8), dtype)


def test_k9_forward_works_and_training_is_refused_early(pkg):
    """The gate kernel is generic in k; the weight-gradient kernel is instantiated for k = 1, 3, 5, 7 -- so a k=9 model
    runs forward / inference and is refused a TRAINING workspace with a NintError naming the layer (not a shape error
    out of the first backward())."""
    from oracle import convlstm_oracle as O
    params = O.synth_params(3, [8], [9], 1, seed=7)
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.standard_normal((2, 2, 3, 12, 20)).astype(np.float32))
    net = pkg.ConvLSTM(3, [8], [9], 1).cuda()
    net.load_state_dict(params)
    with torch.no_grad():
        pred = net(X.cuda()).cpu()
    np.testing.assert_allclose(pred.numpy(), O.convlstm_forward(X, params).numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(pkg.NintError, match="layer 0.*k=9"):
        net(X.cuda())


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(7, [24, 16, 8], [5, 3, 3], 2, 1, 5, 37, 50), (62, [64, 32, 16], [5, 3, 3], 20, 2, 3, 100, 154), (62, [64, 32, 16], [5, 3, 3], 20, 4, 2, 100, 154),
                                   (5, [16, 16], [3, 3], 1, 3, 1, 9, 17), (6, [16, 16, 8, 8], [3, 3, 3, 3], 1, 2, 3, 21, 40),
                                   (6, [8, 8, 8, 8, 8], [3, 3, 3, 3, 3], 1, 1, 3, 12, 20)])
def test_forward_wavefront_as_one_grid_per_step_changes_no_bit(pkg, shape, dtype):
    """nint_seq.wave (include/nint.h): the (t, layer) wavefront of the forward pass, each step's gate launches merged into
    ONE grid (conv_lstm_multi_kernel) -- the same workgroups on the same data -- gives the time-major order's prediction and
    gradients bit for bit (small batch at a ragged grid, the bench grid at B = 2, T = 1 with two layers, four layers = the
    merged kernels' limit, five layers = one launch per layer again)."""
    from nasa_niswan_amd import engine
    C, hidden, ks, out, B, T, H, W = shape
    torch.manual_seed(3)
    X = torch.randn(B, T, C, H, W, device="cuda")
    wgt = torch.randn(B, out, H, W, device="cuda")
    res = {}
    old = engine.FORCE_WAVE
    try:
        for wave in (0, 1):
            engine.FORCE_WAVE = wave
            torch.manual_seed(4)
            net = pkg.ConvLSTM(C, hidden, ks, len(hidden), out_channels=out, compute_dtype=dtype).cuda()
            for rep in range(2):                  # (the second pass reuses the workspace)
                net.zero_grad()
                pred = net(X)
                (pred * wgt).sum().backward()
            torch.cuda.synchronize()
            res[wave] = [pred.detach().clone()] + [p.grad.clone() for p in net.parameters()]
            assert {ws.seq.wave for pool in net._engine(X.device).pool.values() for ws in pool} == {wave}
    finally:
        engine.FORCE_WAVE = old
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_forward_wavefront_rule_and_fallback(pkg):
    """The engine's own rule (FORCE_WAVE None) turns the merged grids on for B = 1 at the bench grid (nint_seq.wave = 5: the forward
    wavefront on the layers' own tiles = the time-major order's launches, bit for bit) and the forward wavefront alone, EVERY layer on 8-row tiles, for B = 4
    and B = 8 (wave = 4, round 4: a two-workgroups-per-CU grid anyway; its backward half is the test below).  That form equals the time-major order bit for bit
    with the tile height pinned to 8, and the default time-major order (4-row tiles for the narrow layers: their four K-slice
    partials are summed in another order) to f32 rounding."""
    from nasa_niswan_amd import engine
    assert engine.FORCE_WAVE is None
    torch.manual_seed(5)
    net = pkg.ConvLSTM(62, [64, 32, 16], [5, 3, 3], 3, out_channels=20, compute_dtype="bf16").cuda()
    X1, X4, X8 = (torch.randn(b, 2, 62, 100, 154, device="cuda") for b in (1, 4, 8))
    with torch.no_grad():
        p1, p4, p8 = net(X1), net(X4), net(X8)
    assert {ws.B: ws.seq.wave for pool in net._engine(p1.device).pool.values() for ws in pool} == {1: 5, 4: 4, 8: 4}
    engine.FORCE_WAVE = 0
    try:
        with torch.no_grad():
            q1, q4, r8 = net(X1), net(X4), net(X8)
    finally:
        engine.FORCE_WAVE = None
    assert torch.equal(p1, q1)
    for b, p, r in ((4, p4, q4), (8, p8, r8)):
        rel = float((p - r).norm() / r.norm())
        print(f"  B = {b}: merged 8-row grid vs time-major with 4-row narrow tiles, rel-L2 {rel:.2e}")
        assert rel <= 1e-3
    engine.FORCE_WAVE, engine.FORCE_TILE_ROWS = 0, 8
    try:
        torch.manual_seed(5)
        net8 = pkg.ConvLSTM(62, [64, 32, 16], [5, 3, 3], 3, out_channels=20, compute_dtype="bf16").cuda()
        net8.load_state_dict(net.state_dict())
        with torch.no_grad():
            s4, s8 = net8(X4), net8(X8)
    finally:
        engine.FORCE_WAVE, engine.FORCE_TILE_ROWS = None, 0
    assert torch.equal(p4, s4) and torch.equal(p8, s8)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("waves", [(2, 4), (1, 5)])
@pytest.mark.parametrize("shape", [(7, [24, 16, 8], [5, 3, 3], 2, 1, 5, 37, 50), (62, [64, 32, 16], [5, 3, 3], 20, 4, 3, 100, 154),
                                   (5, [16, 16], [3, 3], 1, 3, 2, 9, 17), (6, [16, 16, 8, 8], [3, 3, 3, 3], 1, 2, 3, 21, 40),
                                   (62, [64, 32, 16], [5, 3, 3], 20, 8, 2, 100, 154), (62, [64, 32, 16], [5, 3, 3], 20, 1, 3, 100, 154)])
def test_dgrad_pair_as_one_grid(pkg, shape, dtype, waves):
    """nint_seq.wave = 4: the bottom layer's dgrad of time u+1 and layer 1's dgrad of time u as ONE grid; each stores its piece of
    the bottom layer's d/dh (the bottom layer's own into the idle split-K scratch) and the pointwise backward adds them.  f32: the
    same f32 sum as the read-modify-write of the time-major order -> every gradient bit for bit.  bf16: the pieces are rounded
    separately -> prediction and the gradients of layers >= 1 bit for bit, layer 0's and the input's to bf16 rounding (measured
    4.5e-4 / 3e-3 relative at the bench shape).  In a stack of three or more with a fused top layer the bottom layer's pointwise
    backward also rides with the top layer's fused step of the next BPTT step (the same arithmetic).  Reference: wave = 2 (the same
    forward pass) for wave = 4, wave = 1 for wave = 5.  With x.requires_grad (dx rides on the bottom dgrad) and on the second pass
    over a reused workspace."""
    from nasa_niswan_amd import engine
    C, hidden, ks, out, B, T, H, W = shape
    torch.manual_seed(3)
    X0 = torch.randn(B, T, C, H, W, device="cuda")
    wgt = torch.randn(B, out, H, W, device="cuda")
    res = {}
    old = engine.FORCE_WAVE
    try:
        for wave in waves:
            engine.FORCE_WAVE = wave
            torch.manual_seed(4)
            net = pkg.ConvLSTM(C, hidden, ks, len(hidden), out_channels=out, compute_dtype=dtype).cuda()
            for rep in range(2):
                net.zero_grad()
                X = X0.clone().requires_grad_(True)
                pred = net(X)
                (pred * wgt).sum().backward()
            torch.cuda.synchronize()
            res[wave] = [("pred", pred.detach().clone()), ("dX", X.grad.clone())] + [(k, p.grad.clone()) for k, p in net.named_parameters()]
            assert {ws.seq.wave for pool in net._engine(X.device).pool.values() for ws in pool} == {wave}
    finally:
        engine.FORCE_WAVE = old
    for (k, a), (_, b) in zip(res[waves[0]], res[waves[1]]):
        if dtype == "f32" or not (k == "dX" or k.startswith("layers.0.")):
            assert torch.equal(a, b), k
        else:
            rel = float((a - b).norm() / (a.norm() + 1e-30))
            print(f"  {k}: rel-L2 {rel:.2e}")
            assert rel <= 1e-2, (k, rel)
