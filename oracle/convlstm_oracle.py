"""CPU restatement of the reference ConvLSTM hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines it follows.  The restatement is
functional (tensors in, tensors out, parameters as a dict keyed exactly like the
reference ``state_dict``) so that it can be run in fp32 or fp64 and so that the
per-kernel pieces of the backward pass can be checked one at a time.

Parameters dict (reference model.py:235-251, test.ipynb:4698-4699):
    layers.{i}.conv.weight  (4*Ch_i, Cin_i+Ch_i, k_i, k_i)   in-ch = [x..., h...], out-ch = [i,f,g,o]
    layers.{i}.conv.bias    (4*Ch_i,)
    conv.weight             (out, Ch_last, 1, 1)
    conv.bias               (out,)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "init_params", "num_layers_of", "cell_forward", "cell_forward_stash", "cell_backward",
    "convlstm_forward", "crop_pred", "loss_mse_l1", "loss_mse_l1_grad", "adam_step_numpy",
    "train_step", "steplr", "r2_score_np", "head_forward", "synth_params", "synth_batch",
]


# --------------------------------------------------------------------------- params
def init_params(input_channels: int, hidden_channels: Sequence[int], kernel_size: Sequence[int],
                num_layers: int, out_channels: int = 1, seed: int = 0,
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """PyTorch default ``nn.Conv2d`` init under ``torch.manual_seed(seed)``, created in
    the same order as the reference constructor (model.py:241-251; train.py:32,48: seed(0)
    then ``ConvLSTM(...)``, no ``initialize_weights`` for the LSTM family)."""
    assert len(hidden_channels) == num_layers  # model.py:237
    torch.manual_seed(seed)
    params: Dict[str, torch.Tensor] = {}
    cin = input_channels
    for i in range(num_layers):
        ch, k = hidden_channels[i], kernel_size[i]
        conv = torch.nn.Conv2d(cin + ch, 4 * ch, k, padding=k // 2, bias=True)
        params[f"layers.{i}.conv.weight"] = conv.weight.detach().to(dtype).clone()
        params[f"layers.{i}.conv.bias"] = conv.bias.detach().to(dtype).clone()
        cin = ch
    head = torch.nn.Conv2d(hidden_channels[-1], out_channels, kernel_size=1)
    params["conv.weight"] = head.weight.detach().to(dtype).clone()
    params["conv.bias"] = head.bias.detach().to(dtype).clone()
    return params


def num_layers_of(params: Dict[str, torch.Tensor]) -> int:
    n = 0
    while f"layers.{n}.conv.weight" in params:
        n += 1
    return n


# --------------------------------------------------------------------------- cell
def cell_forward(x, h, c, weight, bias, preact: Optional[list] = None):
    """reference model.py:216-231.  x (B,Cin,H,W); h,c (B,Ch,H,W).
    ``preact``: a list that receives the pre-activation gate tensor (B,4Ch,H,W), out-channel order [i,f,g,o], with
    ``retain_grad()`` set -- after ``backward()`` its ``.grad`` is the dG of SURVEY.md section 8 a-5, which the tests
    compare with the dG slab the BPTT kernels store."""
    ch = h.shape[1]
    k = weight.shape[-1]
    combined = torch.cat([x, h], dim=1)                          # model.py:219
    gates = F.conv2d(combined, weight, bias, padding=k // 2)     # model.py:220 (zeros padding)
    if preact is not None:
        if gates.requires_grad:
            gates.retain_grad()
        preact.append(gates)
    gi, gf, gg, go = torch.split(gates, ch, dim=1)               # model.py:221
    gi = torch.sigmoid(gi)                                       # model.py:223
    gf = torch.sigmoid(gf)                                       # model.py:224
    gg = torch.tanh(gg)                                          # model.py:225
    go = torch.sigmoid(go)                                       # model.py:226
    c_new = c * gf + gi * gg                                     # model.py:228
    h_new = go * torch.tanh(c_new)                               # model.py:229
    return h_new, c_new


def cell_forward_stash(x, h, c, weight, bias):
    """Same as :func:`cell_forward` but also returns the post-activation gates that the
    backward pass needs (what autograd saves implicitly in the reference)."""
    ch = h.shape[1]
    k = weight.shape[-1]
    gates = F.conv2d(torch.cat([x, h], dim=1), weight, bias, padding=k // 2)
    gi, gf, gg, go = torch.split(gates, ch, dim=1)
    gi, gf, gg, go = torch.sigmoid(gi), torch.sigmoid(gf), torch.tanh(gg), torch.sigmoid(go)
    c_new = c * gf + gi * gg
    h_new = go * torch.tanh(c_new)
    return h_new, c_new, (gi, gf, gg, go)


def cell_backward(x, h_prev, c_prev, weight, gates, c_new, dh, dc):
    """Explicit BPTT step of one cell (the autograd backward of model.py:216-231,
    SURVEY.md section 8 a-5).  Returns (dx, dh_prev, dc_prev, dW, db, dG)."""
    gi, gf, gg, go = gates
    k = weight.shape[-1]
    cin = x.shape[1]
    tc = torch.tanh(c_new)
    d_o = dh * tc
    dct = dc + dh * go * (1 - tc * tc)
    d_i = dct * gg
    d_f = dct * c_prev
    d_g = dct * gi
    dc_prev = dct * gf
    dG = torch.cat([d_i * gi * (1 - gi), d_f * gf * (1 - gf), d_g * (1 - gg * gg), d_o * go * (1 - go)], dim=1)
    combined = torch.cat([x, h_prev], dim=1)
    db = dG.sum(dim=(0, 2, 3))
    dW = torch.nn.grad.conv2d_weight(combined, weight.shape, dG, padding=k // 2)
    dcomb = torch.nn.grad.conv2d_input(combined.shape, weight, dG, padding=k // 2)
    return dcomb[:, :cin], dcomb[:, cin:], dc_prev, dW, db, dG


# --------------------------------------------------------------------------- sequence model
def head_forward(h, w_head, b_head):
    """1x1 bottleneck conv on the last layer's final hidden state (model.py:251,274)."""
    return F.conv2d(h, w_head, b_head)


def convlstm_forward(x, params, return_states: bool = False, return_sequence: bool = False,
                     h0: Optional[List[torch.Tensor]] = None, c0: Optional[List[torch.Tensor]] = None,
                     preact: Optional[Dict[Tuple[int, int], torch.Tensor]] = None):
    """reference model.py:253-274.  x (B,T,C,H,W) -> (B,out,H,W).

    ``return_sequence`` restates the commented-out variant (model.py:264,272,274) that
    the analysis notebook was run with (test.ipynb:273): per-step head outputs
    concatenated on the channel axis.  ``preact``: a dict that receives {(layer, t): pre-activation gates}
    (see :func:`cell_forward`)."""
    L = num_layers_of(params)
    B, T, _, H, W = x.shape
    hs, cs = [], []
    for i in range(L):
        ch = params[f"layers.{i}.conv.bias"].shape[0] // 4
        hs.append(torch.zeros(B, ch, H, W, dtype=x.dtype) if h0 is None else h0[i])   # model.py:260
        cs.append(torch.zeros(B, ch, H, W, dtype=x.dtype) if c0 is None else c0[i])   # model.py:261
    outs = []
    for t in range(T):                                           # model.py:265
        x_t = x[:, t]                                            # model.py:266
        for i in range(L):                                       # model.py:267
            pa = [] if preact is not None else None
            h, c = cell_forward(x_t, hs[i], cs[i], params[f"layers.{i}.conv.weight"],
                                params[f"layers.{i}.conv.bias"], pa)  # model.py:269
            if pa:
                preact[(i, t)] = pa[0]
            hs[i], cs[i] = h, c                                  # model.py:270
            x_t = h                                              # model.py:271
        if return_sequence:
            outs.append(head_forward(hs[-1], params["conv.weight"], params["conv.bias"]))  # model.py:272
    pred = head_forward(hs[-1], params["conv.weight"], params["conv.bias"])               # model.py:274
    if return_sequence:
        return pred, torch.cat(outs, dim=1)
    if return_states:
        return pred, hs, cs
    return pred


# --------------------------------------------------------------------------- loss
def crop_pred(out, halo: Tuple[int, int], grid: Tuple[int, int]):
    """train.py:102 / utils.py:71: ``pred[:, :, 5:5+90, 5:5+144].squeeze()`` generalised to
    (halo_y, halo_x) and (H, W); the reference hard-codes halo 5 and 90x144."""
    hy, hx = halo
    H, W = grid
    return out[:, :, hy:hy + H, hx:hx + W]


def loss_mse_l1(y, pred):
    """train.py:74-75,105: ``MSELoss()(y, pred) + L1Loss()(y, pred)``, mean reduction.
    (input/target are swapped in the reference; the value is symmetric.)"""
    d = y - pred
    return (d * d).mean() + d.abs().mean()


def loss_mse_l1_grad(y, pred):
    """d loss / d pred for :func:`loss_mse_l1` with torch's sign(0)=0 convention."""
    n = pred.numel()
    d = pred - y
    return (2.0 * d + torch.sign(d)) / n


# --------------------------------------------------------------------------- optimiser
def adam_step_numpy(p, g, m, v, step: int, lr: float, betas=(0.5, 0.999), eps: float = 1e-8):
    """torch.optim.Adam single-tensor update (train.py:71,110; defaults eps=1e-8,
    weight_decay=0, amsgrad=False), same operation order as torch's ``_single_tensor_adam``:
        m = lerp(m, g, 1-b1) (torch's two-branch lerp); v = b2*v + (1-b2)*g*g
        denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m/denom
    ``step`` is the 1-based step count AFTER this update.  Arrays are float32 numpy; the
    bias-correction scalars are computed in Python floats (as torch does)."""
    b1, b2 = betas
    p = np.asarray(p, dtype=np.float32)
    g = np.asarray(g, dtype=np.float32)
    m = np.asarray(m, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    w1 = np.float32(1 - b1)
    if w1 < 0.5:       # torch lerp: a + w*(b-a) for w < 0.5, else b - (b-a)*(1-w)
        m = (m + w1 * (g - m)).astype(np.float32)
    else:
        m = (g - (g - m) * (np.float32(1) - w1)).astype(np.float32)
    v = (v * np.float32(b2) + (np.float32(1 - b2) * g) * g).astype(np.float32)   # addcmul: (value*t1)*t2
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)).astype(np.float32)
    p = (p + (np.float32(-step_size) * m) / denom).astype(np.float32)             # addcdiv: (value*t1)/t2
    return p, m, v


def steplr(lr0: float, epoch: int, step_size: int, gamma: float) -> float:
    """optim.lr_scheduler.StepLR closed form (train.py:72,120): lr after ``epoch`` calls
    of ``scheduler.step()``."""
    return lr0 * gamma ** (epoch // step_size)


def r2_score_np(y_true, y_pred) -> float:
    """sklearn.metrics.r2_score on flattened arrays (train.py:114, utils.py:73)."""
    y_true = np.asarray(y_true, dtype=np.float64).ravel()
    y_pred = np.asarray(y_pred, dtype=np.float64).ravel()
    ss_res = ((y_true - y_pred) ** 2).sum()
    ss_tot = ((y_true - y_true.mean()) ** 2).sum()
    return float(1.0 - ss_res / ss_tot)


# --------------------------------------------------------------------------- one fit-loop step
def train_step(params: Dict[str, torch.Tensor], opt_state: Optional[dict], X, y,
               lr: float, betas=(0.5, 0.999), halo=(0, 0)):
    """One batch of the reference fit loop (train.py:96-110): forward, crop, squeeze,
    MSE+L1, zero_grad, backward (torch CPU autograd), Adam step.

    ``params`` are updated out-of-place; returns (new_params, new_opt_state, loss, pred, grads).
    ``opt_state`` = {"step": int, "m": {k: tensor}, "v": {k: tensor}} or None."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = convlstm_forward(X, leaf)
    H, W = y.shape[-2], y.shape[-1]
    pred = crop_pred(out, halo, (H, W)).squeeze()                # train.py:102
    loss = loss_mse_l1(y, pred)                                  # train.py:105
    loss.backward()                                              # train.py:109
    grads = {k: v.grad.detach().clone() for k, v in leaf.items()}
    if opt_state is None:
        opt_state = {"step": 0,
                     "m": {k: torch.zeros_like(v) for k, v in params.items()},
                     "v": {k: torch.zeros_like(v) for k, v in params.items()}}
    step = opt_state["step"] + 1
    new_p, new_m, new_v = {}, {}, {}
    for k in params:                                             # train.py:110
        p, m, v = adam_step_numpy(params[k].numpy(), grads[k].numpy(), opt_state["m"][k].numpy(),
                                  opt_state["v"][k].numpy(), step, lr, betas)
        new_p[k], new_m[k], new_v[k] = torch.from_numpy(p), torch.from_numpy(m), torch.from_numpy(v)
    return new_p, {"step": step, "m": new_m, "v": new_v}, float(loss.detach()), pred.detach(), grads


# --------------------------------------------------------------------------- seeded synthetic params
def synth_params(input_channels: int, hidden_channels: Sequence[int], kernel_size: Sequence[int],
                 num_layers: int, out_channels: int = 1, seed: int = 0,
                 dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Parameters drawn from ``numpy.random.default_rng(seed)`` with the same bounds as
    PyTorch's default Conv2d init (U(+-1/sqrt(fan_in))).  numpy's Generator stream is stable
    across versions, so fixtures can store just the seed instead of megabytes of weights."""
    assert len(hidden_channels) == num_layers
    rng = np.random.default_rng(seed)
    params: Dict[str, torch.Tensor] = {}
    cin = input_channels
    for i in range(num_layers):
        ch, k = hidden_channels[i], kernel_size[i]
        bound = 1.0 / math.sqrt((cin + ch) * k * k)
        params[f"layers.{i}.conv.weight"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(4 * ch, cin + ch, k, k)).astype(np.float32)).to(dtype)
        params[f"layers.{i}.conv.bias"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(4 * ch,)).astype(np.float32)).to(dtype)
        cin = ch
    bound = 1.0 / math.sqrt(hidden_channels[-1])
    params["conv.weight"] = torch.from_numpy(
        rng.uniform(-bound, bound, size=(out_channels, hidden_channels[-1], 1, 1)).astype(np.float32)).to(dtype)
    params["conv.bias"] = torch.from_numpy(rng.uniform(-bound, bound, size=(out_channels,)).astype(np.float32)).to(dtype)
    return params


def synth_batch(B: int, T: int, C: int, Hp: int, Wp: int, grid: Tuple[int, int], seed: int = 0):
    """Seeded synthetic (X, y): z-scored inputs are ~N(0,1) (dataset.py:528)."""
    rng = np.random.default_rng(seed + 1000)
    X = torch.from_numpy(rng.standard_normal((B, T, C, Hp, Wp)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((B, grid[0], grid[1])).astype(np.float32))
    return X, y
