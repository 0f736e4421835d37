"""Flat parameter / gradient bucket and the fused Adam step (reference train.py:71,108-110).

``FlatParams`` re-homes every parameter of a module into ONE contiguous f32 buffer (and its
gradient into a second one).  That single bucket is what the HIP Adam kernel updates in one
launch and what RCCL all-reduces in one call under data parallelism: every ConvLSTM weight is
used at every time step, so all gradients become final together at the end of BPTT and there
is nothing to overlap bucket-by-bucket.

``FusedAdam`` subclasses ``torch.optim.Optimizer`` so that ``StepLR`` (train.py:72) drives it
unchanged and ``state_dict()`` is the torch Adam format (``state[i] = {step, exp_avg,
exp_avg_sq}``, ``param_groups``) -- reference checkpoints (utils.py:23-50) interchange."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class FlatParams:
    def __init__(self, module: torch.nn.Module):
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters()]
        if not self.params:
            raise ValueError("module has no parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.data = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1).float())
            p.data = self.data[off:off + k].view(p.shape)        # the parameter now aliases the bucket
            p.grad = self.grad[off:off + k].view(p.shape)        # and so does its .grad
            self.offsets.append(off)
            off += k
        self.numel = n

    def grad_view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.grad[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)

    def is_intact(self) -> bool:
        """False once something (``.to()``, ``load_state_dict(assign=True)``...) re-allocated a parameter."""
        base = self.data.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (eps 1e-8, no weight decay, no amsgrad) in one HIP launch."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat = flat
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(flat.params, defaults)
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self._step = 0
        self._bind_state()

    def _bind_state(self):
        for p, off in zip(self.flat.params, self.flat.offsets):
            k = p.numel()
            self.state[p] = {"step": torch.tensor(float(self._step)),
                             "exp_avg": self.exp_avg[off:off + k].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape)}

    def zero_grad(self, set_to_none: bool = False):
        # one memset of the bucket; the .grad views stay aliased (reference call site: train.py:108).
        # The fused trainer never needs it: its backward overwrites the bucket.
        self.flat.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        if not self.flat.is_intact():
            raise RuntimeError("a parameter was re-allocated after FlatParams was built; rebuild the optimizer")
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        check(_lib.load().nint_adam_flat(ptr(self.flat.data), ptr(self.flat.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                         self.flat.numel, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                         self._step, float(grad_scale), stream_ptr()), "nint_adam_flat")
        for p in self.flat.params:
            self.state[p]["step"] = torch.tensor(float(self._step))
        return None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # torch-format dict (reference utils.py:42)
        steps = []
        for p, off in zip(self.flat.params, self.flat.offsets):
            st = self.state.get(p, {})
            k = p.numel()
            if "exp_avg" in st:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st["step"])))
        self._step = max(steps) if steps else 0
        self._bind_state()
