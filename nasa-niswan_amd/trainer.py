"""One batch of the reference fit loop (train.py:92-114) as a fixed sequence of HIP launches.

    X.cuda(), y.cuda()                      -> inputs already on device
    pred = generator(X)                     -> nint_pack_btchw + nint_seq_fwd + nint_head_fwd
    pred[:, :, 5:95, 5:149].squeeze()       -> crop folded into the loss kernel (index math)
    MSELoss(y,pred) + L1Loss(y,pred)        -> nint_loss_mse_l1_crop (also emits d loss/d pred)
    zero_grad; loss.backward()              -> nint_head_bwd + nint_seq_bwd (grads overwrite the bucket)
    [DDP]                                   -> ONE RCCL all-reduce of the flat gradient bucket
    optimizer.step()                        -> nint_adam_flat (1/world folded in)
    loss.item(); r2_score(...cpu())         -> device-side accumulators, read once per epoch

No autograd graph, no per-kernel Python, no host synchronisation inside the step."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import NINT_LOSS_SCRATCH_FLOATS, NINT_LOSS_STATS, check, ptr, stream_ptr
from .model import ConvLSTM
from .optim import FlatParams, FusedAdam


class FusedTrainer:
    def __init__(self, model: ConvLSTM, lr: float = 1e-3, betas=(0.5, 0.999), eps: float = 1e-8,
                 halo: Tuple[int, int] = (5, 5), process_group=None, distributed: Optional[bool] = None,
                 overlap_allreduce: bool = False):
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.NintError("FusedTrainer needs the model on the MI355X (cuda)")
        self.model = model
        self.device = dev
        self.flat = FlatParams(model)
        self.optimizer = FusedAdam(self.flat, lr=lr, betas=betas, eps=eps)
        self.halo = tuple(halo)
        self.lib = _lib.load()
        self.scratch = torch.zeros(NINT_LOSS_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        # [0..4] pooled: sum d^2, sum |d|, sum y, sum y^2, n; [5..7] per call: sum loss, sum r2_score, calls
        self.stats = torch.zeros(NINT_LOSS_STATS, dtype=torch.float64, device=dev)
        self._dpred = None
        self._probe = (None, 0)
        self._marks = None      # diagnostic: a list that step() fills with HIP events (start, after forward, after head/loss, after BPTT, end)
        import torch.distributed as dist
        self.dist = dist
        self.pg = process_group
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if distributed else 1
        self.distributed = bool(distributed)      # True also for a 1-rank group (exercises the collective path)
        # Data-parallel exchange in two pieces (worth it for the strong-scaling shapes, where the step is ~1.5 ms and
        # the one all-reduce at its end ~5 % of it): the bucket without layer 0's slice is reduced UNDER layer 0's weight gradient
        # (nint_seq.bwd_parts), layer 0's slice after it.  Two all-reduces of disjoint slices: the same sums.
        self.overlap_allreduce = bool(overlap_allreduce)
        L = model.num_layers
        self._dW = [self.flat.grad_view(2 * l) for l in range(L)]
        self._db = [self.flat.grad_view(2 * l + 1) for l in range(L)]
        self._dw_head = self.flat.grad_view(2 * L)
        self._db_head = self.flat.grad_view(2 * L + 1)
        if self.distributed:
            # identical initial weights on every rank (the reference seeds identically, utils.py:77-88)
            dist.broadcast(self.flat.data, src=0, group=process_group)

    def set_probe(self, buf: Optional[torch.Tensor], mask: int = 0):
        """In-step timing probes (nint_seq.probe, include/nint.h): `buf` = int64/uint64 device tensor of 2*slots words that
        the following step() calls fill with {tag, 100 MHz timestamp} pairs around every launch whose kind is in `mask`;
        None switches them off.  Diagnostic: bench.py prices the kernels INSIDE the step with it."""
        self._probe = (buf, int(mask) if buf is not None else 0)

    # ------------------------------------------------------------------ pieces
    def forward_loss(self, X: torch.Tensor, y: torch.Tensor, train: bool = True):
        m = self.model
        eng = m._engine(self.device)
        B, T, _, H, W = X.shape
        ws = eng.acquire(B, T, H, W, train, False)
        wb_w = [c.conv.weight for c in m.layers]
        wb_b = [c.conv.bias for c in m.layers]
        eng.pack_weights(wb_w, wb_b)
        eng.forward(ws, X)
        pred = eng.head_forward(ws, m.conv.weight, m.conv.bias)
        O = pred.shape[1]
        Hc, Wc = y.shape[-2], y.shape[-1]
        yv = y.detach().float().contiguous()
        assert yv.numel() == B * O * Hc * Wc, "target must be (B,[O,]Hc,Wc)"
        dpred = None
        if train:
            if self._dpred is None or self._dpred.shape != pred.shape:
                self._dpred = torch.empty_like(pred)
            dpred = self._dpred
        check(self.lib.nint_loss_mse_l1_crop(ptr(pred), ptr(yv), ptr(dpred), ptr(self.scratch), ptr(self.stats),
                                             B, O, H, W, self.halo[0], self.halo[1], Hc, Wc, stream_ptr()),
              "nint_loss_mse_l1_crop")
        return eng, ws, pred, dpred

    def step(self, X: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One optimisation step; returns the loss as a 0-dim DEVICE tensor (no sync).  X: (B,T,C,Hp,Wp) f32 or a
        dataset.SlabBatch."""
        m = self.model
        eng = m._engine(self.device)
        B, T, _, H, W = X.shape
        L = m.num_layers
        ws = eng.acquire(B, T, H, W, True, False)
        pb, pm = self._probe
        ws.seq.probe = pb.data_ptr() if pb is not None else None
        ws.seq.probe_mask, ws.seq.probe_slots = pm, (pb.numel() // 2 if pb is not None else 0)
        mark = self._mark
        mark()
        eng.pack_weights([c.conv.weight for c in m.layers], [c.conv.bias for c in m.layers])
        eng.forward(ws, X)
        mark()
        O = m.conv.weight.shape[0]
        Hc, Wc = y.shape[-2], y.shape[-1]
        yv = y.detach().float().contiguous()
        assert yv.numel() == B * O * Hc * Wc, "target must be (B,[O,]Hc,Wc)"
        if self._dpred is None or self._dpred.shape != (B, O, H, W):
            self._dpred = torch.empty(B, O, H, W, dtype=torch.float32, device=self.device)
        dpred = self._dpred
        # head forward + crop + loss + dpred + dL/dh in one pass; the prediction itself is never materialised
        fused = eng.head_loss_fused(ws, m.conv.weight, m.conv.bias, yv, dpred, self.scratch, self.stats, self.halo, Hc, Wc)
        if not fused:
            pred = eng.head_forward(ws, m.conv.weight, m.conv.bias)
            check(self.lib.nint_loss_mse_l1_crop(ptr(pred), ptr(yv), ptr(dpred), ptr(self.scratch), ptr(self.stats),
                                                 B, O, H, W, self.halo[0], self.halo[1], Hc, Wc, stream_ptr()),
                  "nint_loss_mse_l1_crop")
        eng.head_backward(ws, m.conv.weight, dpred, dw_out=self._dw_head, db_out=self._db_head, write_dh=not fused)
        mark()
        if self.distributed and self.overlap_allreduce and L > 1 and pb is None:
            n0 = self.flat.offsets[2]              # layers.0.conv.weight + .bias lead the bucket (module parameter order)
            eng.backward(ws, False, zero_state_grads=range(L), dW_out=self._dW, db_out=self._db, parts=1)
            work = self.dist.all_reduce(self.flat.grad[n0:], op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True)
            eng.backward(ws, False, zero_state_grads=range(L), dW_out=self._dW, db_out=self._db, parts=2)
            mark()
            eng.release(ws)
            self.dist.all_reduce(self.flat.grad[:n0], op=self.dist.ReduceOp.SUM, group=self.pg)
            work.wait()                            # (stream-level for RCCL: the step stays asynchronous to the host)
            self.optimizer.step(grad_scale=1.0 / self.world)
            mark()
            return self.scratch[0]
        eng.backward(ws, False, zero_state_grads=range(L), dW_out=self._dW, db_out=self._db)
        mark()
        eng.release(ws)
        if self.distributed:
            self.dist.all_reduce(self.flat.grad, op=self.dist.ReduceOp.SUM, group=self.pg)   # RCCL over xGMI
        self.optimizer.step(grad_scale=1.0 / self.world)
        mark()
        return self.scratch[0]

    def _mark(self):
        """bench.py --phase-events: one HIP event per phase boundary of a step (nothing is recorded unless asked for)"""
        if self._marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self._marks.append(ev)

    @torch.no_grad()
    def evaluate(self, X: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Forward + loss only (val_loop, utils.py:52-75); accumulates the R2 statistics."""
        eng, ws, pred, _ = self.forward_loss(X, y, False)
        eng.release(ws)
        return pred

    # ------------------------------------------------------------------ epoch statistics
    def reset_stats(self):
        self.stats.zero_()

    def epoch_stats(self, pooled: bool = False):
        """One device->host read per epoch.  Returns what the reference logs over everything accumulated since
        reset_stats(): (mean of the per-batch losses, mean of the per-batch sklearn ``r2_score``) -- train.py:113-117;
        for the validation loop at batch size 1 that is the mean of per-sample R2 (utils.py:73-75).  Under DDP
        the sums are all-reduced first, so the means run over every rank's batches.
        ``pooled=True`` appends the pooled R2 over all elements seen (1 - sum d^2 / sum (y - mean y)^2)."""
        s = self.stats.clone()
        if self.world > 1:
            self.dist.all_reduce(s, op=self.dist.ReduceOp.SUM, group=self.pg)
        s2, s1, sy, syy, n, sl, sr2, calls = (float(v) for v in s.cpu())
        if calls == 0:
            return (float("nan"),) * (3 if pooled else 2)
        out = (sl / calls, sr2 / calls)
        if pooled:
            ss_tot = syy - sy * sy / n
            out += (1.0 - s2 / ss_tot if ss_tot > 0 else float("nan"),)
        return out
