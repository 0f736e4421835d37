"""The CPU oracle against the golden vectors produced by the reference itself
(oracle/make_goldens.py) and the reference's own known answers.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import convlstm_oracle as O
from oracle import preproc_oracle as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


CELLS = [(4, 8, 3), (5, 16, 5), (16, 8, 3), (5, 64, 5), (64, 32, 3), (32, 16, 3)]


def cell_inputs(g):
    cin, ch, k, seed = int(g["cin"]), int(g["ch"]), int(g["k"]), int(g["seed"])
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    p = O.synth_params(cin, [ch], [k], 1, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    h = torch.from_numpy((0.5 * rng.standard_normal((B, ch, H, W))).astype(np.float32))
    c = torch.from_numpy(rng.standard_normal((B, ch, H, W)).astype(np.float32))
    dh = torch.from_numpy(rng.standard_normal((B, ch, H, W)).astype(np.float32))
    dc = torch.from_numpy(rng.standard_normal((B, ch, H, W)).astype(np.float32))
    return p["layers.0.conv.weight"], p["layers.0.conv.bias"], x, h, c, dh, dc


@pytest.mark.parametrize("cin,ch,k", CELLS)
def test_cell_forward_backward_vs_reference(cin, ch, k):
    g = load(f"cell_{cin}_{ch}_{k}.npz")
    w, b, x, h, c, dh, dc = cell_inputs(g)
    h1, c1, gates = O.cell_forward_stash(x, h, c, w, b)
    np.testing.assert_allclose(h1.numpy(), g["h_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c1.numpy(), g["c_out"], rtol=1e-5, atol=1e-6)
    dx, dhp, dcp, dW, db, _ = O.cell_backward(x, h, c, w, gates, c1, dh, dc)
    np.testing.assert_allclose(dx.numpy(), g["dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dhp.numpy(), g["dh_prev"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dcp.numpy(), g["dc_prev"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(db.numpy(), g["db"], rtol=1e-4, atol=1e-4)
    assert abs(float(dW.norm()) - float(g["dW_l2"])) <= 1e-4 * float(g["dW_l2"])
    if g["dW"].size:
        np.testing.assert_allclose(dW.numpy(), g["dW"], rtol=1e-4, atol=1e-4)


def test_cfg0_fit_loop_vs_reference():
    g = load("cfg0_train.npz")
    params = {k[len("params0."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("params0.")}
    # train.py:32,48: seed(0) + default init == oracle.init_params
    init = O.init_params(4, [8], [3], 1, seed=0)
    for k in params:
        np.testing.assert_array_equal(init[k].numpy(), params[k].numpy())
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    state = None
    for step in (1, 2, 3):
        params, state, loss, pred, grads = O.train_step(params, state, X, y, lr=float(g["lr"]), betas=tuple(g["betas"]))
        assert abs(loss - float(g[f"loss{step}"])) < 1e-6
        if step == 1:
            np.testing.assert_allclose(pred.numpy(), g["pred_full"].squeeze(), rtol=1e-5, atol=1e-6)
            for k in grads:
                np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], rtol=1e-4, atol=1e-7)
        if step in (1, 3):
            for k in params:
                np.testing.assert_allclose(params[k].numpy(), g[f"params{step}." + k], rtol=1e-5, atol=2e-7)


def test_small3_fit_loop_vs_reference():
    g = load("small3_train.npz")
    hidden, ks = [int(v) for v in g["hidden"]], [int(v) for v in g["ks"]]
    params = O.synth_params(int(g["C"]), hidden, ks, 3, seed=int(g["seed"]))
    X, y = O.synth_batch(int(g["B"]), int(g["T"]), int(g["C"]), int(g["Hp"]), int(g["Wp"]),
                         tuple(int(v) for v in g["grid"]), seed=int(g["seed"]))
    state = None
    for step in (1, 2, 3):
        params, state, loss, pred, grads = O.train_step(params, state, X, y, lr=float(g["lr"]),
                                                        betas=tuple(g["betas"]), halo=tuple(int(v) for v in g["halo"]))
        assert abs(loss - float(g[f"loss{step}"])) < 2e-6
        if step == 1:
            for k in grads:
                np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], rtol=1e-4, atol=1e-7)
        if step in (1, 3):
            for k in params:
                np.testing.assert_allclose(params[k].numpy(), g[f"params{step}." + k], rtol=1e-5, atol=2e-6)


def test_refsize_vs_reference_and_known_answers():
    g = load("refsize_train.npz")
    hidden, ks = [int(v) for v in g["hidden"]], [int(v) for v in g["ks"]]
    params = O.synth_params(5, hidden, ks, 3, seed=int(g["seed"]))
    # test.ipynb:4698-4699: parameter-count known answer
    assert [v.numel() for v in params.values()] == [441600, 256, 110592, 128, 27648, 64, 16, 1]
    assert sum(v.numel() for v in params.values()) == 580305
    X, y = O.synth_batch(1, 2, 5, 18, 22, (8, 12), seed=int(g["seed"]))
    out = O.convlstm_forward(X, params)
    np.testing.assert_allclose(out.numpy(), g["pred_full"], rtol=1e-5, atol=1e-6)
    new_p, _, loss, _, grads = O.train_step(params, None, X, y, lr=1e-3, halo=(5, 5))
    assert abs(loss - float(g["loss1"])) < 1e-6
    np.testing.assert_allclose(grads["layers.0.conv.weight"].numpy()[::16, ::8], g["g0_slice"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(grads["layers.1.conv.weight"].numpy()[::8, ::8], g["g1_slice"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(grads["layers.2.conv.weight"].numpy(), g["grad.layers.2.conv.weight"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(new_p["conv.weight"].numpy(), g["p1_head_w"], rtol=1e-5, atol=1e-6)


def test_adam_restatement_vs_torch():
    g = load("adam.npz")
    p = g["p0"]
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for i, grad in enumerate(g["grads"], 1):
        p, m, v = O.adam_step_numpy(p, grad, m, v, i, float(g["lr"]), tuple(g["betas"]))
        np.testing.assert_allclose(p, g["p_after"][i - 1], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(m, g["m"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(v, g["v"], rtol=1e-6, atol=1e-12)


def test_steplr_and_r2():
    assert O.steplr(1e-3, 0, 10, 0.5) == 1e-3
    assert abs(O.steplr(1e-3, 30, 10, 0.5) - 1.25e-4) < 1e-12       # test.ipynb:200
    from sklearn.metrics import r2_score
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(500), rng.standard_normal(500)
    assert abs(O.r2_score_np(a, b) - r2_score(a, b)) < 1e-12


def test_loss_grad_matches_autograd():
    rng = np.random.default_rng(1)
    y = torch.from_numpy(rng.standard_normal((2, 9, 7)).astype(np.float32))
    p = torch.from_numpy(rng.standard_normal((2, 9, 7)).astype(np.float32)).requires_grad_(True)
    O.loss_mse_l1(y, p).backward()
    np.testing.assert_allclose(O.loss_mse_l1_grad(y, p.detach()).numpy(), p.grad.numpy(), rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------ preproc
def test_padding_3d_matches_notebook_golden():
    # dataset_config.ipynb:484-502
    got = P.padding_data_3d(np.arange(25).reshape(1, 5, 5), (13, 13))
    assert got.shape == (1, 13, 13)
    np.testing.assert_array_equal(got, P.NOTEBOOK_13x13)


def test_padding_4d_shapes_and_quirk():
    g = load("pad4d_quirk.npz")
    out = P.padding_data_4d(g["small_in"], (11, 12))
    np.testing.assert_array_equal(out, g["small_out"])
    np.testing.assert_array_equal(P.padding_data_4d(g["small_in"], (11, 12), "reflect"), g["small_out_reflect"])
    x = np.random.default_rng(int(g["x_seed"])).standard_normal((3, 5, 90, 144)).astype(np.float32)
    y = P.padding_data_4d(x, (100, 154))
    assert y.shape == (3, 5, 100, 154)          # dataset_config.ipynb:712: X (10,5,100,154)
    assert abs(float(y.astype(np.float64).sum()) - float(g["out_checksum"])) < 1e-6
    # interior is the cyclic-padded field; the halo rows come from channel C-1-c, unflipped (dataset.py:96)
    np.testing.assert_array_equal(y[:, :, 5:95, 5:149], x)
    np.testing.assert_array_equal(y[:, 0, 0:5, 5:149], x[:, 4, 1:6, :])
    np.testing.assert_array_equal(y[:, 1, 95:100, 5:149], x[:, 3, 84:89, :])
    # the reflect mode is a true mirror
    z = P.padding_data_4d(x, (100, 154), "reflect")
    np.testing.assert_array_equal(z[:, 2, 0:5, 5:149], x[:, 2, 5:0:-1, :])
    # oversize padding raises like the reference (dataset.py:80,98)
    with pytest.raises(AttributeError):
        P.cyclic_pad(np.zeros((1, 1, 4, 4)), 20)


def test_fuse_levels_reference_case():
    rng = np.random.default_rng(2)
    u, v, w, pr, s = (rng.standard_normal((3, 6, 8)).astype(np.float32) for _ in range(5))
    x = P.fuse_levels(u, v, w, pr, s)
    np.testing.assert_array_equal(x, np.stack([u, v, w, pr, s], axis=1))      # dataset.py:526
    u3 = rng.standard_normal((3, 4, 6, 8)).astype(np.float32)
    x = P.fuse_levels(u3, u3 + 1, u3 + 2, pr, s)
    assert x.shape == (3, 14, 6, 8)
    np.testing.assert_array_equal(x[:, 4:8], u3 + 1)
    np.testing.assert_array_equal(x[:, 13], s)
