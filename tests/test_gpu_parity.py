"""GPU parity tests: the HIP path (through the C ABI / drop-in modules) against the CPU oracle
and the golden vectors produced by the reference's own model.py.

Tolerances (stated once, used everywhere):
  f32 mode : outputs rtol 1e-4 / atol 1e-5; gradients max-abs error <= 1e-3 * max|grad| (+1e-6)
             (different summation order over K <= 4750 and over B*H*W*T for wgrad; the reference's
             own f32-vs-f64 noise floor is 2e-6 relative)
  bf16 mode: rel-L2 error <= 2e-2 on outputs and <= 2e-2 on gradients against the f32 oracle
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="module")
def N():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as pkg
    pkg.load_library()
    return pkg


def dev(a):
    return torch.as_tensor(np.asarray(a)).cuda()


def maxerr(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a - b).max()), float(np.abs(b).max())


def rel_l2(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def assert_close_out(a, b, what):
    e, m = maxerr(a, b)
    print(f"  {what}: max abs err {e:.3e} (ref max {m:.3e})")
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5, err_msg=what)


def assert_close_grad(a, b, what):
    e, m = maxerr(a, b)
    print(f"  {what}: max abs err {e:.3e} (ref max {m:.3e})")
    assert e <= 1e-3 * m + 1e-6, f"{what}: {e} vs tolerance {1e-3 * m + 1e-6}"


def assert_bf16(a, b, what, tol):
    r = rel_l2(a, b)
    print(f"  {what}: rel-L2 {r:.3e}")
    assert r <= tol, f"{what}: rel-L2 {r} > {tol}"


# ------------------------------------------------------------------ hardware assumptions
def test_selftest_mfma_and_transposed_read(N):
    lib = N.load_library()
    out = torch.zeros(4096, device="cuda")
    assert lib.nint_selftest(C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    lane = np.arange(64)
    for base, name in ((0, "16x16x32_bf16"), (1024, "16x16x4_f32")):
        d = o[base:base + 256].reshape(64, 4)
        # C/D map: column = lane&15, row = 4*(lane>>4) + reg ; D[m][n] = (m+1)*(32+n)
        for r in range(4):
            m = 4 * (lane >> 4) + r
            n = lane & 15
            np.testing.assert_array_equal(d[:, r], (m + 1) * (32 + n), err_msg=name)
    tr = o[2048:2048 + 256].reshape(64, 4)
    # lane i of a 16-lane group g receives column i of rows 4g..4g+3 (one row per element)
    g, i16 = lane >> 4, lane & 15
    for e in range(4):
        np.testing.assert_array_equal(tr[:, e], 64 * (4 * g + e) + i16)


def test_device_is_gfx950(N):
    lib = N.load_library()
    ncu, lds, wave = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    assert lib.nint_device_info(C.byref(ncu), C.byref(lds), C.byref(wave), name, 64) == 0
    print("  device:", name.value.decode(), "CUs", ncu.value, "LDS/CU", lds.value, "wave", wave.value)
    assert name.value.decode().startswith("gfx950") and wave.value == 64 and ncu.value == 256


# ------------------------------------------------------------------ cell step (model.py:216-231)
CELLS = [(4, 8, 3), (5, 16, 5), (16, 8, 3), (5, 64, 5), (64, 32, 3), (32, 16, 3)]


def cell_case(cin, ch, k):
    from oracle import convlstm_oracle as O
    g = load(f"cell_{cin}_{ch}_{k}.npz")
    seed = int(g["seed"])
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    p = O.synth_params(cin, [ch], [k], 1, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    h = (0.5 * rng.standard_normal((B, ch, H, W))).astype(np.float32)
    c = rng.standard_normal((B, ch, H, W)).astype(np.float32)
    dh = rng.standard_normal((B, ch, H, W)).astype(np.float32)
    dc = rng.standard_normal((B, ch, H, W)).astype(np.float32)
    return g, p, x, h, c, dh, dc


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cin,ch,k", CELLS)
def test_cell_forward_backward(N, cin, ch, k, dtype):
    g, p, x, h, c, dh, dc = cell_case(cin, ch, k)
    cell = N.ConvLSTMCell(cin, ch, k, compute_dtype=dtype).cuda()
    cell.load_state_dict({"conv.weight": p["layers.0.conv.weight"], "conv.bias": p["layers.0.conv.bias"]})
    xt, ht, ct = (dev(a).requires_grad_(True) for a in (x, h, c))
    h1, c1 = cell(xt, (ht, ct))
    ((h1 * dev(dh)).sum() + (c1 * dev(dc)).sum()).backward()
    torch.cuda.synchronize()
    if dtype == "f32":
        assert_close_out(h1, g["h_out"], "h_out")
        assert_close_out(c1, g["c_out"], "c_out")
        assert_close_grad(xt.grad, g["dx"], "dx")
        assert_close_grad(ht.grad, g["dh_prev"], "dh_prev")
        assert_close_grad(ct.grad, g["dc_prev"], "dc_prev")
        assert_close_grad(cell.conv.bias.grad, g["db"], "db")
        l2 = float(cell.conv.weight.grad.norm())
        print(f"  dW l2 {l2:.6e} vs {float(g['dW_l2']):.6e}")
        assert abs(l2 - float(g["dW_l2"])) <= 1e-3 * float(g["dW_l2"])
        if g["dW"].size:
            assert_close_grad(cell.conv.weight.grad, g["dW"], "dW")
    else:
        assert_bf16(h1, g["h_out"], "h_out", 2e-2)
        assert_bf16(c1, g["c_out"], "c_out", 2e-2)
        assert_bf16(xt.grad, g["dx"], "dx", 2e-2)
        assert_bf16(ht.grad, g["dh_prev"], "dh_prev", 2e-2)
        assert_bf16(ct.grad, g["dc_prev"], "dc_prev", 2e-2)
        assert_bf16(cell.conv.bias.grad, g["db"], "db", 2e-2)
        if g["dW"].size:
            assert_bf16(cell.conv.weight.grad, g["dW"], "dW", 2e-2)


# ------------------------------------------------------------------ whole model + fit-loop step
def _train_case(name):
    from oracle import convlstm_oracle as O
    g = load(name)
    if name == "cfg0_train.npz":
        params = {k[len("params0."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("params0.")}
        cfg = dict(C=4, hidden=[8], ks=[3], halo=(0, 0))
        X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    else:
        hidden, ks = [int(v) for v in g["hidden"]], [int(v) for v in g["ks"]]
        cfg = dict(C=int(g["C"]), hidden=hidden, ks=ks, halo=tuple(int(v) for v in g["halo"]))
        params = O.synth_params(cfg["C"], hidden, ks, len(hidden), seed=int(g["seed"]))
        X, y = O.synth_batch(int(g["B"]), int(g["T"]), cfg["C"], int(g["Hp"]), int(g["Wp"]),
                             tuple(int(v) for v in g["grid"]), seed=int(g["seed"]))
    return g, cfg, params, X, y


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["cfg0_train.npz", "small3_train.npz", "refsize_train.npz"])
def test_model_forward_backward_vs_reference(N, name, dtype):
    from oracle import convlstm_oracle as O
    g, cfg, params, X, y = _train_case(name)
    net = N.ConvLSTM(cfg["C"], cfg["hidden"], cfg["ks"], len(cfg["hidden"]), compute_dtype=dtype).cuda()
    net.load_state_dict(params)
    Xd, yd = X.cuda(), y.cuda()
    out = net(Xd)
    hy, hx = cfg["halo"]
    H, W = y.shape[-2], y.shape[-1]
    pred = out[:, :, hy:hy + H, hx:hx + W].squeeze()                    # train.py:102
    loss = torch.nn.functional.mse_loss(yd, pred) + torch.nn.functional.l1_loss(yd, pred) \
        if pred.shape == yd.shape else ((yd - pred) ** 2).mean() + (yd - pred).abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    # oracle on the same inputs (the goldens hold only slices for the big tensors)
    _, _, oloss, _, ograds = O.train_step(params, None, X, y, lr=1e-3, halo=cfg["halo"])
    print(f"  loss {float(loss):.7f} oracle {oloss:.7f} golden {float(g['loss1']):.7f}")
    grads = dict(net.named_parameters())
    if dtype == "f32":
        assert_close_out(out, g["pred_full"], "pred")
        assert abs(float(loss) - float(g["loss1"])) < 1e-5
        for k in ograds:
            assert_close_grad(grads[k].grad, ograds[k], "grad." + k)
            if ("grad." + k) in g.files:
                assert_close_grad(grads[k].grad, g["grad." + k], "golden grad." + k)
    else:
        assert_bf16(out, g["pred_full"], "pred", 2e-2)
        assert abs(float(loss) - float(g["loss1"])) < 2e-2 * abs(float(g["loss1"]))
        for k in ograds:
            assert_bf16(grads[k].grad, ograds[k], "grad." + k, 2e-2)


def test_inference_matches_training_forward_and_sequence_head(N):
    from oracle import convlstm_oracle as O
    g, cfg, params, X, y = _train_case("small3_train.npz")
    net = N.ConvLSTM(cfg["C"], cfg["hidden"], cfg["ks"], 3, return_sequence=True).cuda().eval()
    net.load_state_dict(params)
    with torch.no_grad():
        pred, seq = net(X.cuda())
    opred, oseq = O.convlstm_forward(X, params, return_sequence=True)
    assert_close_out(pred, opred, "pred (no_grad)")
    assert_close_out(seq, oseq, "per-step head outputs")
    assert seq.shape == (X.shape[0], X.shape[1], X.shape[3], X.shape[4])


def test_input_gradient(N):
    from oracle import convlstm_oracle as O
    g, cfg, params, X, y = _train_case("small3_train.npz")
    net = N.ConvLSTM(cfg["C"], cfg["hidden"], cfg["ks"], 3).cuda()
    net.load_state_dict(params)
    Xd = X.cuda().requires_grad_(True)
    w = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 1, 20, 28)).astype(np.float32))
    (net(Xd) * w.cuda()).sum().backward()
    Xo = X.clone().requires_grad_(True)
    (O.convlstm_forward(Xo, params) * w).sum().backward()
    assert_close_grad(Xd.grad, Xo.grad, "dL/dX")
