#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation itself.

Run in the build container only (``python oracle/make_goldens.py``): it imports
``/root/reference/model.py`` (needs only torch), loads seeded parameters into the
reference ``ConvLSTMCell`` / ``ConvLSTM`` modules, runs forward / backward /
``torch.optim.Adam`` on the CPU and stores the inputs (or their seeds) and outputs as
small fixtures.  The reference source never leaves the container: fixtures are data only.

The pre-processing module of the reference (dataset.py) cannot be imported here
(xarray / torchvision are absent and cannot be installed), so its goldens are
  * the 13x13 padding matrix the reference's own notebook prints
    (dataset_config.ipynb:484-502), transcribed in oracle/preproc_oracle.py, and
  * for the 4-D RNN pad (no golden anywhere in the reference): the numpy restatement of
    dataset.py:67-98 -- marked "parity unpinned" in the fixture's ``note`` field.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import model as ref_model  # noqa: E402  (the reference's model.py)

from oracle import convlstm_oracle as O  # noqa: E402
from oracle import preproc_oracle as P  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def npd(d):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ (a) single cell steps
def golden_cells():
    cases = [(4, 8, 3), (5, 16, 5), (16, 8, 3), (5, 64, 5), (64, 32, 3), (32, 16, 3)]
    for (cin, ch, k) in cases:
        seed = 100 + cin * 7 + ch
        p = O.synth_params(cin, [ch], [k], 1, seed=seed)
        rng = np.random.default_rng(seed + 1)
        B, H, W = 2, 12, 16
        x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
        h = torch.from_numpy((0.5 * rng.standard_normal((B, ch, H, W))).astype(np.float32))
        c = torch.from_numpy(rng.standard_normal((B, ch, H, W)).astype(np.float32))
        cell = ref_model.ConvLSTMCell(cin, ch, k)
        cell.load_state_dict({"conv.weight": p["layers.0.conv.weight"], "conv.bias": p["layers.0.conv.bias"]})
        x.requires_grad_(True); h.requires_grad_(True); c.requires_grad_(True)
        h1, c1 = cell(x, (h, c))
        dh = torch.from_numpy(rng.standard_normal(h1.shape).astype(np.float32))
        dc = torch.from_numpy(rng.standard_normal(c1.shape).astype(np.float32))
        (h1 * dh).sum().backward(retain_graph=True)
        (c1 * dc).sum().backward()
        save(f"cell_{cin}_{ch}_{k}.npz", cin=cin, ch=ch, k=k, seed=seed, B=B, H=H, W=W,
             h_out=h1.detach().numpy(), c_out=c1.detach().numpy(),
             dx=x.grad.numpy(), dh_prev=h.grad.numpy(), dc_prev=c.grad.numpy(),
             dW=cell.conv.weight.grad.numpy().astype(np.float32) if ch <= 16 else np.zeros(0, np.float32),
             dW_l2=float(cell.conv.weight.grad.norm()), dW_sum=float(cell.conv.weight.grad.double().sum()),
             db=cell.conv.bias.grad.numpy())


# ------------------------------------------------------------------ (b) full fit-loop steps
def run_reference_training(cin, hidden, ks, L, B, T, Hp, Wp, halo, grid, lr, betas, nsteps, seed,
                           torch_init: bool):
    if torch_init:
        # exactly what train.py:32,48 does: seed(0) then construct -> PyTorch default init
        torch.manual_seed(seed)
        net = ref_model.ConvLSTM(cin, list(hidden), list(ks), L)
        params0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    else:
        net = ref_model.ConvLSTM(cin, list(hidden), list(ks), L)
        params0 = O.synth_params(cin, hidden, ks, L, seed=seed)
        net.load_state_dict(params0)
    X, y = O.synth_batch(B, T, cin, Hp, Wp, grid, seed=seed)
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=betas)          # train.py:71
    l1, l2 = torch.nn.MSELoss(), torch.nn.L1Loss()                          # train.py:74-75
    out = {}
    for step in range(1, nsteps + 1):
        pred_full = net(X)                                                 # train.py:96
        pred = pred_full[:, :, halo[0]:halo[0] + grid[0], halo[1]:halo[1] + grid[1]].squeeze()  # train.py:102
        loss = l1(y, pred) + l2(y, pred)                                   # train.py:105
        opt.zero_grad(); loss.backward(); opt.step()                        # train.py:108-110
        if step == 1:
            out["pred_full"] = pred_full.detach().numpy().copy()
            out["loss1"] = float(loss)
            for k, v in net.named_parameters():
                out["grad." + k] = v.grad.detach().numpy().copy()
        out[f"loss{step}"] = float(loss)
        if step in (1, 3, nsteps):
            for k, v in net.state_dict().items():
                out[f"params{step}." + k] = v.detach().numpy().copy()
    return params0, X, y, out


def golden_cfg0():
    # BASELINE.json configs[0]: 1-layer ConvLSTM, 32x32, 4 in-channels, seq_len=4, batch=2
    params0, X, y, out = run_reference_training(4, [8], [3], 1, 2, 4, 32, 32, (0, 0), (32, 32),
                                                lr=1e-4, betas=(0.5, 0.999), nsteps=3, seed=0, torch_init=True)
    save("cfg0_train.npz", X=X.numpy(), y=y.numpy(), lr=1e-4, betas=np.array([0.5, 0.999]),
         **{"params0." + k: v.numpy() for k, v in params0.items()}, **out)


def golden_small3():
    # 3 layers with the reference's kernel pattern (5,3,3), halo-5 crop like train.py:102
    hidden, ks = [16, 8, 8], [5, 3, 3]
    params0, X, y, out = run_reference_training(5, hidden, ks, 3, 2, 3, 20, 28, (5, 5), (10, 18),
                                                lr=1e-3, betas=(0.5, 0.999), nsteps=3, seed=7, torch_init=False)
    save("small3_train.npz", seed=7, hidden=np.array(hidden), ks=np.array(ks), C=5, B=2, T=3, Hp=20, Wp=28,
         halo=np.array([5, 5]), grid=np.array([10, 18]), lr=1e-3, betas=np.array([0.5, 0.999]), **out)


def golden_refsize():
    # the reference-size stack (launcher.sh:19-21) on a small padded grid; params from a seed
    hidden, ks = [64, 32, 16], [5, 3, 3]
    params0, X, y, out = run_reference_training(5, hidden, ks, 3, 1, 2, 18, 22, (5, 5), (8, 12),
                                                lr=1e-3, betas=(0.5, 0.999), nsteps=1, seed=11, torch_init=False)
    keep = {k: v for k, v in out.items() if not k.startswith("params") and not k.startswith("grad.layers.0.conv.weight")
            and not k.startswith("grad.layers.1.conv.weight")}
    g0 = out["grad.layers.0.conv.weight"]; g1 = out["grad.layers.1.conv.weight"]
    save("refsize_train.npz", seed=11, hidden=np.array(hidden), ks=np.array(ks), C=5, B=1, T=2, Hp=18, Wp=22,
         halo=np.array([5, 5]), grid=np.array([8, 12]), lr=1e-3,
         g0_l2=float(np.linalg.norm(g0)), g0_slice=g0[::16, ::8].copy(),
         g1_l2=float(np.linalg.norm(g1)), g1_slice=g1[::8, ::8].copy(),
         p1_head_w=out["params1.conv.weight"], p1_b0=out["params1.layers.0.conv.bias"], **keep)
    # known answers from the reference notebook (test.ipynb:4698-4699)
    net = ref_model.ConvLSTM(5, hidden, ks, 3)
    counts = [p.numel() for p in net.parameters()]
    assert counts == [441600, 256, 110592, 128, 27648, 64, 16, 1] and sum(counts) == 580305
    # model.py:282-295 smoke: output shape (2,1,100,154) -- checked at T=1 to keep it quick
    assert tuple(net(torch.zeros(2, 1, 5, 100, 154)).shape) == (2, 1, 100, 154)


# ------------------------------------------------------------------ (c) padding
def golden_padding():
    got = P.padding_data_3d(np.arange(25).reshape(1, 5, 5), (13, 13))
    assert np.array_equal(got, P.NOTEBOOK_13x13), "restatement disagrees with dataset_config.ipynb:484-502"
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 5, 90, 144)).astype(np.float32)
    save("pad4d_quirk.npz", note="parity unpinned: numpy restatement of dataset.py:67-98 (no reference golden)",
         x_seed=5, out_checksum=float(P.padding_data_4d(x, (100, 154)).astype(np.float64).sum()),
         small_in=np.arange(2 * 3 * 7 * 8, dtype=np.float32).reshape(2, 3, 7, 8),
         small_out=P.padding_data_4d(np.arange(2 * 3 * 7 * 8, dtype=np.float32).reshape(2, 3, 7, 8), (11, 12)),
         small_out_reflect=P.padding_data_4d(np.arange(2 * 3 * 7 * 8, dtype=np.float32).reshape(2, 3, 7, 8), (11, 12), "reflect"))


# ------------------------------------------------------------------ (d) Adam vs torch.optim.Adam
def golden_adam():
    rng = np.random.default_rng(3)
    p = rng.standard_normal(1000).astype(np.float32)
    gs = [rng.standard_normal(1000).astype(np.float32) * s for s in (1.0, 1e-3, 10.0, 1e-6)]
    tp = torch.nn.Parameter(torch.from_numpy(p.copy()))
    opt = torch.optim.Adam([tp], lr=1e-3, betas=(0.5, 0.999))
    outs = []
    for g in gs:
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        outs.append(tp.detach().numpy().copy())
    st = opt.state[tp]
    save("adam.npz", p0=p, grads=np.stack(gs), p_after=np.stack(outs), m=st["exp_avg"].numpy(),
         v=st["exp_avg_sq"].numpy(), lr=1e-3, betas=np.array([0.5, 0.999]))


if __name__ == "__main__":
    golden_cells()
    golden_cfg0()
    golden_small3()
    golden_refsize()
    golden_padding()
    golden_adam()
