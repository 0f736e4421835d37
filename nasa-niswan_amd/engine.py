"""Host-side engine: owns the device workspaces (allocated through torch's caching allocator,
sized for 288 GB of HBM: every time step's slabs stay resident) and drives the C-ABI HIP
library for one ConvLSTM forward / BPTT.  No arithmetic happens in Python or in torch ops."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import NINT_BF16, NINT_F32, NintGeom, NintLayer, NintSeq, check, ptr, stream_ptr

# 0: the library picks the gate / dgrad pixel-tile height per launch shape; 4 or 8 forces it for every layer of
# engines built afterwards (nint_layer.tile_rows) -- how the tests run both heights on every shape
FORCE_TILE_ROWS = 0
# Thin first-layer inputs (the reference's Conv2d(5+64 -> 256, k=5), model.py:207-211) are fed HORIZONTALLY FOLDED
# when that lowers the number of MFMA K-steps (nint_xfold_pays): 5 x-steps instead of 25.  (Any channel count and grid width
# packs: row tiles up to 160 KiB go through the LDS-tiled pack kernel, wider ones through the per-element one.)  False keeps the plain
# channel-padded layout for engines built afterwards (tests run both).
XFOLD = True
# nint_layer.wide of engines built afterwards (weight-gradient kernel family): 0 = the library's choice, 1 = always the 4-wave
# 64-column kernel, 2 = the 8-wave 128-column kernel wherever it is instantiated (tests run both against each other)
FORCE_WIDE = 0
FORCE_WAVE = None        # merged grids (nint_seq.wave): None = by batch size (SeqEngine._set_wave), 0 = never, 1 = forward wavefront + backward pair, 2 = forward wavefront only (8-row tiles), 4 = 2 + the BPTT pairs (dgrad 0 + dgrad 1, pointwise 0 + the top layer's fused step), 5 = the forward pass of 1 + the BPTT pairs of 4
WAVE_TILES_PER_CU = 1.5   # ... wave = 5 (the layers' own tiles) while 2 * (8-row pixel tiles of the batch) < WAVE_TILES_PER_CU * CUs: B = 1 at 100 x 154; wave = 4 above
FUSE_BWD = 0          # nint_seq.fuse_bwd of new workspaces: 0 = per layer, 1 = never fused, 2 = every layer fused (tests run all three)

DTYPES = {"f32": NINT_F32, "fp32": NINT_F32, "float32": NINT_F32, "bf16": NINT_BF16, "bfloat16": NINT_BF16}


def dtype_code(d) -> int:
    if isinstance(d, int):
        return d
    if isinstance(d, torch.dtype):
        return NINT_BF16 if d == torch.bfloat16 else NINT_F32
    return DTYPES[str(d).lower()]


def _rup(a: int, b: int) -> int:
    return (a + b - 1) // b * b


@dataclass
class LayerCfg:
    Cx: int
    Ch: int
    k: int
    xfold: bool = False      # x source horizontally folded (thin first-layer inputs; set by SeqEngine, nint_layer.xfold)

    def padded(self, kc: int) -> Tuple[int, int, int]:
        return _rup(self.k * self.Cx if self.xfold else self.Cx, kc), _rup(self.Ch, 16), _rup(self.Ch, kc)


class Workspace:
    """All slabs of one (B, T, H, W) problem.  Halo slabs are zero-initialised once; kernels only
    ever write their interior, so the zero padding of nn.Conv2d (model.py:207-211) is physical."""

    def __init__(self, eng: "SeqEngine", B: int, T: int, H: int, W: int, train: bool, has_init: bool):
        self.key = (B, T, H, W, train, has_init)
        self.B, self.T, self.H, self.W, self.train = B, T, H, W, train
        self.in_use = False
        dev, es, kc = eng.device, eng.es, eng.kc
        lib = _lib.load()
        self.g = NintGeom()
        check(lib.nint_geom_make(C.byref(self.g), H, W, eng.P), "nint_geom_make")
        g = self.g
        halo_px, comp_px = g.Hh * g.Wh, H * W
        u8 = dict(dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        Cxp0 = eng.cfgs[0].padded(kc)[0]
        self.xs = torch.zeros(T * B * halo_px * Cxp0 * es, **u8)
        self.h, self.c, self.gates, self.dG, self.dh, self.dc = [], [], [], [], [], []
        for cfg in eng.cfgs:
            Cxp, Ch16, Chp = cfg.padded(kc)
            self.h.append(torch.zeros((T + 1) * B * halo_px * Chp * es, **u8))
            self.c.append(torch.zeros((T + 1) * B * comp_px * Chp, **f32))
            if train:
                self.gates.append(torch.empty(T * B * comp_px * 4 * Ch16 * es, **u8))
                self.dG.append(torch.zeros(T * B * halo_px * 4 * Ch16 * es, **u8))
                self.dh.append(torch.zeros(B * comp_px * Chp * es, **u8))      # ET: transient gradient, read once per step
                self.dc.append(torch.zeros(B * comp_px * Chp, **f32))
        self.dx = None
        self.Cxp0 = Cxp0
        self.seq = NintSeq()
        s = self.seq
        s.dtype, s.B, s.T, s.L = eng.dt, B, T, len(eng.cfgs)
        s.need_dx, s.has_init_state, s.n_cu = 0, int(has_init), eng.n_cu
        s.fuse_bwd = FUSE_BWD
        s.g = g
        s.xs = self.xs.data_ptr()
        for l in range(len(eng.cfgs)):
            s.h[l] = self.h[l].data_ptr()
            s.c[l] = self.c[l].data_ptr()
            if train:
                s.gates[l] = self.gates[l].data_ptr()
                s.dG[l] = self.dG[l].data_ptr()
                s.dh[l] = self.dh[l].data_ptr()
                s.dc[l] = self.dc[l].data_ptr()
        if train:
            s.wg_partial = eng.wg_partial.data_ptr()
            s.wg_partial_bytes = eng.wg_partial.numel() * 4

    # byte offsets of slab (t) inside the per-layer stacks
    def h_view(self, eng, l: int, slot: int) -> int:
        Chp = eng.cfgs[l].padded(eng.kc)[2]
        return self.h[l].data_ptr() + slot * self.B * self.g.Hh * self.g.Wh * Chp * eng.es

    def c_view(self, eng, l: int, slot: int) -> int:
        Chp = eng.cfgs[l].padded(eng.kc)[2]
        return self.c[l].data_ptr() + slot * self.B * self.H * self.W * Chp * 4


class SeqEngine:
    def __init__(self, cfgs: Sequence[LayerCfg], dtype="f32", device="cuda"):
        if len(cfgs) < 1 or len(cfgs) > _lib.NINT_MAX_LAYERS:
            raise ValueError("1..8 layers supported")
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NintError("the ConvLSTM hot path runs on the MI355X only (device must be cuda); "
                                 "there is no CPU fallback")
        self.dt = dtype_code(dtype)
        self.es = 2 if self.dt == NINT_BF16 else 4
        self.kc = self.lib.nint_kc(self.dt)
        self.cfgs = [LayerCfg(c.Cx, c.Ch, c.k, bool(l == 0 and XFOLD and self.lib.nint_xfold_pays(c.Cx, c.k, self.dt)))
                     for l, c in enumerate(cfgs)]
        self.P = max(c.k // 2 for c in cfgs)
        for c in cfgs:
            if c.k % 2 == 0:
                raise ValueError("odd kernel sizes only (padding k//2, model.py:204)")
        n_cu = C.c_int(0)
        with torch.cuda.device(self.device):
            check(self.lib.nint_device_info(C.byref(n_cu), None, None, None, 0), "nint_device_info")
        self.n_cu = n_cu.value
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.Wf, self.Wd, self.bias_p, self.layers = [], [], [], []
        wg_bytes = 0
        for cfg in self.cfgs:
            nbytes = self.lib.nint_packed_weight_bytes(cfg.Cx, cfg.Ch, cfg.k, self.dt, int(cfg.xfold))
            Cxp, Ch16, Chp = cfg.padded(self.kc)
            self.Wf.append(torch.zeros(nbytes, **u8))
            self.Wd.append(torch.zeros(nbytes, **u8))
            self.bias_p.append(torch.zeros(4 * Ch16, dtype=torch.float32, device=self.device))
            ly = NintLayer()
            ly.Cx, ly.Cxp, ly.Ch, ly.Ch16, ly.Chp, ly.k = cfg.Cx, Cxp, cfg.Ch, Ch16, Chp, cfg.k
            ly.xfold = int(cfg.xfold)
            ly.Wf, ly.Wd, ly.bias_p = self.Wf[-1].data_ptr(), self.Wd[-1].data_ptr(), self.bias_p[-1].data_ptr()
            self.layers.append(ly)
            ly.tile_rows = FORCE_TILE_ROWS
            ly.wide = FORCE_WIDE
            # the layers' split-K slabs sit side by side in one workspace: one launch folds them all
            wg_bytes += (self.lib.nint_wgrad_workspace_bytes(C.byref(ly), self.dt, self.n_cu) + 255) // 256 * 256
        # shapes the weight-gradient kernel has no instantiation for: known NOW, reported at the first training
        # workspace (forward / inference work for every odd k) instead of as a shape error in the first backward()
        self.train_unsupported = self.untrainable_layers(self.cfgs, self.dt, self.n_cu)
        self._wg_bytes = wg_bytes
        self._wg_partial = None
        self.pool: Dict[tuple, List[Workspace]] = {}

    @property
    def wg_partial(self):
        if self._wg_partial is None:
            self._wg_partial = torch.empty(self._wg_bytes // 4 + 16, dtype=torch.float32, device=self.device)
        return self._wg_partial

    # ------------------------------------------------------------------ weights
    def pack_weights(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]):
        """OIHW f32 conv weights of every layer -> MFMA fragment order (fwd + dgrad images) and permuted bias: one launch."""
        L = len(self.cfgs)
        Ws, bs = [], []
        for l, cfg in enumerate(self.cfgs):
            W = weights[l].detach()
            if W.dtype != torch.float32 or not W.is_contiguous():
                W = W.float().contiguous()
            assert tuple(W.shape) == (4 * cfg.Ch, cfg.Cx + cfg.Ch, cfg.k, cfg.k), W.shape
            b = biases[l]
            if b is not None:
                b = b.detach()
                if b.dtype != torch.float32 or not b.is_contiguous():
                    b = b.float().contiguous()
            Ws.append(W); bs.append(b)
        wp = (C.c_void_p * L)(*[W.data_ptr() for W in Ws])
        bp = (C.c_void_p * L)(*[None if b is None else b.data_ptr() for b in bs])
        lys = (NintLayer * L)(*self.layers)
        check(self.lib.nint_pack_weights_layers(wp, bp, lys, L, self.dt, stream_ptr()), "nint_pack_weights_layers")

    @staticmethod
    def untrainable_layers(cfgs: Sequence[LayerCfg], dtype, n_cu: int = 256) -> List[str]:
        """Layers whose weight gradient no kernel instantiation covers (`nint_wgrad_workspace_bytes` == 0):
        kernel sizes other than 1, 3, 5, 7.  Pure host
        arithmetic -- callable without a GPU."""
        lib = _lib.load()
        dt = dtype_code(dtype)
        kc = lib.nint_kc(dt)
        bad = []
        for l, cfg in enumerate(cfgs):
            ly = NintLayer()
            ly.Cx, ly.Ch, ly.k, ly.xfold = cfg.Cx, cfg.Ch, cfg.k, int(cfg.xfold)
            ly.Cxp, ly.Ch16, ly.Chp = cfg.padded(kc)
            if lib.nint_wgrad_workspace_bytes(C.byref(ly), dt, n_cu) == 0:
                bad.append(f"layer {l} (Cin={cfg.Cx}, Ch={cfg.Ch}, k={cfg.k})")
        return bad

    # ------------------------------------------------------------------ workspaces
    def acquire(self, B, T, H, W, train: bool, has_init: bool) -> Workspace:
        if train and self.train_unsupported:
            raise _lib.NintError("training is not supported for " + ", ".join(self.train_unsupported) + ": the weight-gradient "
                                 "kernel is instantiated for kernel sizes 1, 3, 5 and 7 only (the reference accepts any odd k, "
                                 "model.py:204); forward / inference (torch.no_grad()) work for every odd k")
        key = (B, T, H, W, train, has_init)
        for ws in self.pool.setdefault(key, []):
            if not ws.in_use:
                ws.in_use = True
                ws.seq.probe, ws.seq.probe_mask, ws.seq.probe_slots = None, 0, 0     # (a trainer's timing probes do not outlive its step)
                return ws
        ws = Workspace(self, B, T, H, W, train, has_init)
        for l, ly in enumerate(self.layers):
            ws.seq.layer[l] = ly
        self.pool[key].append(ws)
        ws.in_use = True
        return ws

    @staticmethod
    def release(ws: Workspace):
        ws.in_use = False

    # ------------------------------------------------------------------ passes
    def forward(self, ws: Workspace, x: torch.Tensor, h0: Optional[List[torch.Tensor]] = None,
                c0: Optional[List[torch.Tensor]] = None):
        """x (B,T,C,H,W) f32 on the engine's device, or a dataset.SlabBatch of that shape.  Runs model.py:253-271
        (all layers, all steps)."""
        B, T, Cc, H, W = x.shape
        assert (B, T, H, W) == (ws.B, ws.T, ws.H, ws.W) and Cc == self.cfgs[0].Cx
        st = stream_ptr()
        g = C.byref(ws.g)
        self.pack_input(ws, x)
        if h0 is not None:
            for l, cfg in enumerate(self.cfgs):
                Chp = cfg.padded(self.kc)[2]
                hh = h0[l].detach().float().contiguous().view(B, 1, cfg.Ch, H, W)
                check(self.lib.nint_pack_btchw(ptr(hh), C.c_void_p(ws.h_view(self, l, 0)), B, 1, cfg.Ch, Chp, g, self.dt, st),
                      "pack h0")
                cc = c0[l].detach().float().contiguous()
                check(self.lib.nint_pack_compact(ptr(cc), C.c_void_p(ws.c_view(self, l, 0)), B, cfg.Ch, Chp, H, W, NINT_F32, st),
                      "pack c0")
        self._set_wave(ws)
        check(self.lib.nint_seq_fwd(C.byref(ws.seq), st), "nint_seq_fwd")

    def _set_wave(self, ws: Workspace):
        """nint_seq.wave: independent launches as one grid (a forward wavefront step; in BPTT the dgrad launches of layers 0 and 1
        and the bottom pointwise backward with the top layer's fused step).  FORCE_WAVE = None: wave = 5 for the smallest batches
        (B = 1 at 100 x 154: the layers' own tiles), wave = 4 (every forward layer on 8-row tiles) for everything above;
        an int: that mode (include/nint.h)."""
        tiles8 = ws.B * ((ws.W + 15) // 16) * ((ws.H + 7) // 8)
        if FORCE_WAVE is None:
            # round 4, six fresh processes per setting at B = 8 (profiles/r04_c_wave_repeats.txt): the forward wavefront as one
            # grid per step is 2.79-2.83 ms of forward against 2.87-2.94 in every process but one; the backward pair is neutral
            # (4.82-4.93 against 4.84-4.92 ms) with one slow process in six (5.01): that pair was the "bimodal" of round 3.
            # Three fresh processes per mode and batch (profiles/r04_d_wave_small_batches.txt): B = 1: mode 1 647 samples/s
            # against 586 (mode 2); B = 2: 833 against 841; B = 4: 957 against 966; B = 8: see above.
            # ... and from B = 2 the bottom layer's dgrad rides with layer 1's of the next BPTT step (wave = 4: B = 2 / 4 / 8
            # +3.2 / +1.2 / +0.4 % over wave = 2, every fresh-process pair but one of nine positive; B = 1: 601 against 665 of
            # wave = 1: profiles/r04_f_wave4.txt)
            # B = 1: the same BPTT pairs behind the forward wavefront on the layers' own tiles (wave = 5): 661 against 647 (wave = 1)
            # and 622 (wave = 4) samples/s
            # No upper end: B = 12 / 16 / 32 measure +1.0 ... +2.2 % with wave = 4 against the time-major order (forward -5 %,
            # backward neutral: profiles/r04_h_big_batches.txt); rounds 3-4 had stopped the rule at B = 8.
            mode = 5 if 2 * tiles8 < WAVE_TILES_PER_CU * self.n_cu else 4
        else:
            mode = int(FORCE_WAVE)
        ws.seq.wave = mode if len(self.cfgs) > 1 else 0

    def pack_input(self, ws: Workspace, x):
        """Fill the input slab ws.xs (image t*B+b, channels-last ET, folded for thin inputs) from x: a (B,T,C,H,W)
        f32 tensor (nint_pack_btchw[_xfold]) or a dataset.SlabBatch (the preproc kernel writes the slab directly)."""
        B, T, Cc, H, W = x.shape
        if hasattr(x, "fill_slab"):
            x.fill_slab(self, ws)
            return
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        st, g = stream_ptr(), C.byref(ws.g)
        if self.cfgs[0].xfold:
            check(self.lib.nint_pack_btchw_xfold(ptr(x), ptr(ws.xs), B, T, Cc, self.cfgs[0].k, ws.Cxp0, g, self.dt, st),
                  "nint_pack_btchw_xfold")
        else:
            check(self.lib.nint_pack_btchw(ptr(x), ptr(ws.xs), B, T, Cc, ws.Cxp0, g, self.dt, st), "nint_pack_btchw")

    def h_last(self, ws: Workspace, l: int, slot: Optional[int] = None) -> torch.Tensor:
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        slot = ws.T if slot is None else slot
        out = torch.empty(ws.B, cfg.Ch, ws.H, ws.W, dtype=torch.float32, device=self.device)
        check(self.lib.nint_unpack_halo(ptr(ws.h[l]), ptr(out), slot * ws.B, ws.B, cfg.Ch, Chp, C.byref(ws.g), self.dt,
                                        stream_ptr()), "nint_unpack_halo")
        return out

    def c_last(self, ws: Workspace, l: int) -> torch.Tensor:
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        out = torch.empty(ws.B, cfg.Ch, ws.H, ws.W, dtype=torch.float32, device=self.device)
        check(self.lib.nint_unpack_compact(C.c_void_p(ws.c_view(self, l, ws.T)), ptr(out), ws.B, cfg.Ch, Chp, ws.H, ws.W,
                                           NINT_F32, stream_ptr()), "nint_unpack_compact")
        return out

    def head_forward(self, ws: Workspace, w: torch.Tensor, b: Optional[torch.Tensor], slot: Optional[int] = None):
        """model.py:274: 1x1 conv on the last layer's hidden state of time step ``slot-1``."""
        l = len(self.cfgs) - 1
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        O = w.shape[0]
        slot = ws.T if slot is None else slot
        pred = torch.empty(ws.B, O, ws.H, ws.W, dtype=torch.float32, device=self.device)
        w2 = w.detach().float().contiguous()
        b2 = None if b is None else b.detach().float().contiguous()
        check(self.lib.nint_head_fwd(ptr(ws.h[l]), slot * ws.B, ws.B, cfg.Ch, Chp, O, ptr(w2), ptr(b2), ptr(pred),
                                     C.byref(ws.g), self.dt, stream_ptr()), "nint_head_fwd")
        return pred

    def head_loss_fused(self, ws: Workspace, w: torch.Tensor, b: Optional[torch.Tensor], y: torch.Tensor, dpred: torch.Tensor,
                        scratch: torch.Tensor, stats: torch.Tensor, halo, Hc: int, Wc: int) -> bool:
        """Training fast path (nint_head_loss_fused): head forward, crop, MSE+L1 sums, d loss / d pred and dL/dh_{T-1}
        (into ws.dh[-1]) in one pass; `scratch[0]` = loss, `stats` accumulated.  False when the head is wider than the
        fused kernel holds (more than 128 padded channels, or weights + one pixel group's d loss / d pred beyond the LDS) (the caller then takes the three separate launches)."""
        l = len(self.cfgs) - 1
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        O = w.shape[0]
        chv = 32 if Chp <= 32 else (64 if Chp <= 64 else 128)
        if Chp > 128 or (O * chv + min(O, 64) * 64) * 4 + 8192 > 160 * 1024:
            return False
        w2 = w.detach().float().contiguous()
        b2 = None if b is None else b.detach().float().contiguous()
        check(self.lib.nint_head_loss_fused(ptr(ws.h[l]), ws.T * ws.B, ws.B, cfg.Ch, Chp, O, ptr(w2), ptr(b2), ptr(y), ptr(dpred),
                                            ptr(ws.dh[l]), ptr(scratch), ptr(stats), C.byref(ws.g), halo[0], halo[1], Hc, Wc,
                                            self.dt, stream_ptr()), "nint_head_loss_fused")
        return True

    def head_backward(self, ws: Workspace, w: torch.Tensor, dpred: torch.Tensor, dw_out=None, db_out=None, write_dh: bool = True):
        """Writes dL/dh_{T-1} of the last layer into ws.dh[-1] (unless `write_dh` is False: the fused head/loss pass
        already did); returns (dw_head, db_head)."""
        l = len(self.cfgs) - 1
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        O = w.shape[0]
        w2 = w.detach().float().contiguous()
        dp = dpred.detach().float().contiguous()
        dw = dw_out if dw_out is not None else torch.empty(O, cfg.Ch, dtype=torch.float32, device=self.device)
        db = db_out if db_out is not None else torch.empty(O, dtype=torch.float32, device=self.device)
        check(self.lib.nint_head_bwd(ptr(ws.h[l]), ws.T * ws.B, ws.B, cfg.Ch, Chp, O, ptr(w2), ptr(dp),
                                     ptr(ws.dh[l]) if write_dh else None,
                                     ptr(dw), ptr(db), C.byref(ws.g), self.dt, ptr(self.wg_partial),
                                     self.wg_partial.numel() * 4, stream_ptr()), "nint_head_bwd")
        return dw.view(O, cfg.Ch, 1, 1), db

    def backward(self, ws: Workspace, need_dx: bool, zero_state_grads: Sequence[int] = (),
                 dW_out: Optional[Sequence[torch.Tensor]] = None, db_out: Optional[Sequence[torch.Tensor]] = None, parts: int = 0):
        """BPTT of model.py:253-271.  Precondition: ws.dh[l], ws.dc[l] hold dL/dh_{T-1}, dL/dc_{T-1}
        (layers listed in ``zero_state_grads`` start from zero state gradients instead).  Returns ([dW_l], [db_l], dx or None).
        ``dW_out`` / ``db_out``: f32 contiguous destinations (e.g. views of a flat gradient bucket).
        ``parts`` (nint_seq.bwd_parts): 0 = everything; 1 = the BPTT chain + the gradients of layers >= 1; 2 = layer 0's weight /
        bias gradient only, after a parts = 1 call on the same workspace (the trainer starts the all-reduce of the rest of the
        bucket in between)."""
        assert ws.train
        # layers in ``zero_state_grads`` start BPTT from zero dL/dh, dL/dc: flagged, not filled (the first BPTT step
        # then neither reads dc nor accumulates into dh).  The top layer's dh always holds the head's gradient.
        L = len(self.cfgs)
        mask = 0
        for l in zero_state_grads:
            mask |= 1 << (2 * l)
            if l < L - 1:
                mask |= 1 << (2 * l + 1)
        ws.seq.zero_dstate = mask
        dWs, dbs = [], []
        s = ws.seq
        for l, cfg in enumerate(self.cfgs):
            if dW_out is not None:
                dWs.append(dW_out[l])
                dbs.append(db_out[l])
                assert dWs[-1].numel() == 4 * cfg.Ch * (cfg.Cx + cfg.Ch) * cfg.k * cfg.k and dWs[-1].is_contiguous()
            else:
                dWs.append(torch.empty(4 * cfg.Ch, cfg.Cx + cfg.Ch, cfg.k, cfg.k, dtype=torch.float32, device=self.device))
                dbs.append(torch.empty(4 * cfg.Ch, dtype=torch.float32, device=self.device))
            s.dW[l] = dWs[-1].data_ptr()
            s.db[l] = dbs[-1].data_ptr()
        dx = None
        if need_dx:
            dx = torch.empty(ws.T * ws.B * ws.H * ws.W * ws.Cxp0 * self.es, dtype=torch.uint8, device=self.device)   # ET compact
            s.dx = dx.data_ptr()
        s.need_dx = int(need_dx)
        s.bwd_parts = int(parts)
        try:
            check(self.lib.nint_seq_bwd(C.byref(s), stream_ptr()), "nint_seq_bwd")
        finally:
            s.bwd_parts = 0
        s.dx = None
        dx_out = None
        if need_dx:
            # compact [T*B][H][W][Cxp0] -> (B,T,C,H,W)
            C0 = self.cfgs[0].Cx
            tb = torch.empty(ws.T * ws.B, C0, ws.H, ws.W, dtype=torch.float32, device=self.device)
            if self.cfgs[0].xfold:     # gradient of the folded input -> gradient of the input
                check(self.lib.nint_unfold_dx(ptr(dx), ptr(tb), ws.T * ws.B, C0, self.cfgs[0].k, ws.Cxp0, ws.H, ws.W, self.dt,
                                              stream_ptr()), "nint_unfold_dx")
            else:
                check(self.lib.nint_unpack_compact(ptr(dx), ptr(tb), ws.T * ws.B, C0, ws.Cxp0, ws.H, ws.W, self.dt, stream_ptr()),
                      "unpack dx")
            dx_out = tb.view(ws.T, ws.B, C0, ws.H, ws.W).transpose(0, 1).contiguous()
        return dWs, dbs, dx_out

    def state_grads(self, ws: Workspace, l: int):
        """dL/dh_init, dL/dc_init of layer l after backward (has_init_state workspaces)."""
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        dh = torch.empty(ws.B, cfg.Ch, ws.H, ws.W, dtype=torch.float32, device=self.device)
        dc = torch.empty_like(dh)
        st = stream_ptr()
        check(self.lib.nint_unpack_compact(ptr(ws.dh[l]), ptr(dh), ws.B, cfg.Ch, Chp, ws.H, ws.W, self.dt, st), "unpack dh")
        check(self.lib.nint_unpack_compact(ptr(ws.dc[l]), ptr(dc), ws.B, cfg.Ch, Chp, ws.H, ws.W, NINT_F32, st), "unpack dc")
        return dh, dc

    def set_state_grads(self, ws: Workspace, l: int, dh: Optional[torch.Tensor], dc: Optional[torch.Tensor]):
        cfg = self.cfgs[l]
        Chp = cfg.padded(self.kc)[2]
        st = stream_ptr()
        for src, dst, dt in ((dh, ws.dh[l], self.dt), (dc, ws.dc[l], NINT_F32)):
            if src is None:
                dst.zero_()
            else:
                s2 = src.detach().float().contiguous()
                check(self.lib.nint_pack_compact(ptr(s2), ptr(dst), ws.B, cfg.Ch, Chp, ws.H, ws.W, dt, st), "pack dstate")
