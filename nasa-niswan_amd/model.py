"""Drop-in ``ConvLSTMCell`` / ``ConvLSTM`` for the reference's ``model.py`` (lines 196-274).

Same constructor signatures, same module tree and ``state_dict`` keys
(``layers.{i}.conv.weight|bias``, ``conv.weight|bias``: test.ipynb:4698-4699), same return
values -- but ``forward`` runs the MI355X HIP kernels through the C ABI (include/nint.h).
``nn.Conv2d`` objects are kept purely as parameter containers (identical default init under
the same seed, identical checkpoint keys); they are never called.

Extensions are keyword-only with reference-preserving defaults:
    compute_dtype   "f32" (exact-parity mode) or "bf16" (bf16 storage, f32 accumulate)
    out_channels    head width (1 in the reference, L*n_tracers for the level-fused variants)
    return_sequence also return the per-step head outputs (the variant the analysis notebook
                    was run with: model.py:264,272,274 commented code, test.ipynb:273)
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .engine import LayerCfg, SeqEngine, dtype_code

__all__ = ["ConvLSTMCell", "ConvLSTM"]


def _require_cuda(t: torch.Tensor, who: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"{who}: tensors must live on the MI355X (cuda) -- this package has no CPU path; "
                           "move the module and its inputs with .cuda()")


class _Releaser:
    """Returns a workspace to the engine pool when the autograd graph that owns it dies."""

    def __init__(self, ws):
        self.ws = ws
        self._fin = weakref.finalize(self, SeqEngine.release, ws)

    def release(self):
        self._fin()


# ------------------------------------------------------------------------------ autograd glue
class _ConvLSTMFn(torch.autograd.Function):
    """x, head_w, head_b, W_0, b_0, ..., W_{L-1}, b_{L-1} -> pred [, seq]"""

    @staticmethod
    def forward(ctx, module, grad_mode, x, head_w, head_b, *wb):
        eng: SeqEngine = module._engine(x.device)
        B, T, _, H, W = x.shape
        # grad mode is always off inside Function.forward, so the caller samples it
        train = grad_mode and any(ctx.needs_input_grad)
        ws = eng.acquire(B, T, H, W, train, False)
        eng.pack_weights(wb[0::2], wb[1::2])
        eng.forward(ws, x)
        pred = eng.head_forward(ws, head_w, head_b)
        outs = [pred]
        if module.return_sequence:
            if train:
                raise RuntimeError("return_sequence=True is an inference-only extension (test.ipynb:273)")
            outs.append(torch.cat([eng.head_forward(ws, head_w, head_b, slot=t + 1) for t in range(T)], dim=1))
        if train:
            ctx.eng, ctx.ws, ctx.rel = eng, ws, _Releaser(ws)
            ctx.save_for_backward(head_w)
            ctx.x_needs_grad = ctx.needs_input_grad[2]
            ctx.nwb = len(wb)
        else:
            eng.release(ws)
        return tuple(outs) if len(outs) > 1 else pred

    @staticmethod
    def backward(ctx, dpred, *unused):
        eng, ws = ctx.eng, ctx.ws
        (head_w,) = ctx.saved_tensors
        L = len(eng.cfgs)
        dw_head, db_head = eng.head_backward(ws, head_w, dpred)
        dWs, dbs, dx = eng.backward(ws, ctx.x_needs_grad, zero_state_grads=range(L))
        ctx.rel.release()
        grads = []
        for l in range(L):
            grads += [dWs[l], dbs[l]]
        return (None, None, dx, dw_head, db_head, *grads)


class _CellFn(torch.autograd.Function):
    """x, h, c, W, b -> h', c'   (model.py:216-231)"""

    @staticmethod
    def forward(ctx, module, grad_mode, x, h, c, W, b):
        eng: SeqEngine = module._engine(x.device)
        B, _, H, Wd = x.shape
        train = grad_mode and any(ctx.needs_input_grad)
        ws = eng.acquire(B, 1, H, Wd, train, True)
        eng.pack_weights([W], [b])
        eng.forward(ws, x.unsqueeze(1), [h], [c])
        h1, c1 = eng.h_last(ws, 0), eng.c_last(ws, 0)
        if train:
            ctx.eng, ctx.ws, ctx.rel = eng, ws, _Releaser(ws)
            ctx.x_needs_grad = ctx.needs_input_grad[2]
            ctx.has_bias = b is not None
        else:
            eng.release(ws)
        return h1, c1

    @staticmethod
    def backward(ctx, dh1, dc1):
        eng, ws = ctx.eng, ctx.ws
        eng.set_state_grads(ws, 0, dh1, dc1)
        dWs, dbs, dx = eng.backward(ws, ctx.x_needs_grad)
        dh0, dc0 = eng.state_grads(ws, 0)
        ctx.rel.release()
        if dx is not None:
            dx = dx[:, 0]
        return None, None, dx, dh0, dc0, dWs[0], (dbs[0] if ctx.has_bias else None)


class _EngineOwner:
    """Mixin: the per-device engines hold device workspaces and ctypes structures; they are rebuilt on
    demand and must not travel with pickling / copy.deepcopy of the module."""

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"] = {}
        return state


# ------------------------------------------------------------------------------ modules
class ConvLSTMCell(_EngineOwner, nn.Module):
    """reference model.py:196-231"""

    def __init__(self, input_channels, hidden_channels, kernel_size, bias=True, *, compute_dtype="f32"):
        super().__init__()
        self.input_channels = input_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.padding = kernel_size // 2
        self.bias = bias
        # parameter container with the reference's name, shape, layout and default init
        # (out-channel order [i,f,g,o], in-channel order [x..., h...]; model.py:207-211,219-221)
        self.conv = nn.Conv2d(in_channels=self.input_channels + self.hidden_channels,
                              out_channels=4 * self.hidden_channels, kernel_size=self.kernel_size,
                              padding=self.padding, bias=self.bias)
        self.sigmoid = nn.Sigmoid()
        self.tanh = nn.Tanh()
        self.compute_dtype = compute_dtype
        self._engines = {}

    def _engine(self, device) -> SeqEngine:
        key = (str(device), dtype_code(self.compute_dtype))
        if key not in self._engines:
            self._engines[key] = SeqEngine([LayerCfg(self.input_channels, self.hidden_channels, self.kernel_size)],
                                           self.compute_dtype, device)
        return self._engines[key]

    def forward(self, x, hidden_state):
        h, c = hidden_state
        _require_cuda(x, "ConvLSTMCell")
        return _CellFn.apply(self, torch.is_grad_enabled(), x, h, c, self.conv.weight, self.conv.bias)


class ConvLSTM(_EngineOwner, nn.Module):
    """reference model.py:234-274"""

    def __init__(self, input_channels, hidden_channels, kernel_size, num_layers, *, out_channels=1,
                 return_sequence=False, compute_dtype="f32"):
        super().__init__()
        assert len(hidden_channels) == num_layers, 'The length of hidden_channels must be equal to num_layers.'
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        self.layers.append(ConvLSTMCell(input_channels, hidden_channels[0], kernel_size[0], compute_dtype=compute_dtype))
        for i in range(1, num_layers):
            self.layers.append(ConvLSTMCell(hidden_channels[i - 1], hidden_channels[i], kernel_size[i],
                                            compute_dtype=compute_dtype))
        # Bottleneck layer (model.py:251)
        self.conv = nn.Conv2d(hidden_channels[-1], out_channels, kernel_size=1)
        self.return_sequence = return_sequence
        self.compute_dtype = compute_dtype
        self._engines = {}

    def _engine(self, device) -> SeqEngine:
        key = (str(device), dtype_code(self.compute_dtype))
        if key not in self._engines:
            cfgs = [LayerCfg(c.input_channels, c.hidden_channels, c.kernel_size) for c in self.layers]
            self._engines[key] = SeqEngine(cfgs, self.compute_dtype, device)
        return self._engines[key]

    def forward(self, x):
        # x: (batch_size, sequence_length, channels, height, width)  (model.py:254)
        _require_cuda(x, "ConvLSTM")
        wb = []
        for cell in self.layers:
            wb += [cell.conv.weight, cell.conv.bias]
        return _ConvLSTMFn.apply(self, torch.is_grad_enabled(), x, self.conv.weight, self.conv.bias, *wb)
