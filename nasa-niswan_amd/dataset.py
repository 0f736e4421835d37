"""Synthetic, NetCDF-free stand-in for the reference's in-memory RNN dataset
(``E33OMA90D_CRNN``, reference dataset.py:551-637) that feeds the DEVICE pre-processing kernel.

The reference opens the authors' NetCDF files with xarray (absent here, and the data is not
public), so the raw fields are synthesised with the per-variable statistics the reference ships
(``variable_statistics.json`` set1: u, v, omega, prec, bc_src, bc_conc).  Everything after the
file read is reproduced: level selection / fusion order (dataset.py:566-584), z-score with
statistics of the first 70 % of the record (dataset.py:589-596), sliding windows with the
target at the window's last step (dataset.py:598-599,614-616), the 70/10/20 % period split
(dataset.py:601-612) and the cyclic-lon / lat halo pad (dataset.py:61-98) -- the last three on
the GPU in ``nint_preproc_fuse_pad``.

Extension (no reference code): ``levels=L`` keeps u, v, omega at L vertical
levels as 3L level-channels beside the two 2-D fields (C = 3L+2) and predicts the tracer at L
levels; L=1 is the reference.  An ``in_channels`` that is not 3L+2 (BASELINE configs[0]: 4 channels
on a 32x32 grid) synthesises that many generic fields instead of the named ones.

Two device paths feed the model (the whole record stays resident in HBM, a window is a pointer offset):
``device_batch`` materialises the reference's (B,T,C,Hp,Wp) f32 tensor (one launch per batch);
``slab_batch`` returns a handle that the engine asks to write the SAME values straight into its bf16/f32
channels-last input slab -- no f32 intermediate, no pack pass."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

# variable_statistics.json set1 (mean, std): u, v, w(omega), prec, bc_src, bc_conc
STATS = {"u": (0.21191783, 6.5155377), "v": (0.34416693, 5.2940431), "w": (8.225245e-07, 6.2516752e-05),
         "prec": (2.1786141, 7.3012676), "bc_src": (0.19962825, 2.6003716), "bc_conc": (4.9511008, 57.252777)}


class SlabBatch:
    """A batch of windows that has not been materialised: `fill_slab` writes it into an engine workspace's input
    slab with ONE launch of the fuse/z-score/halo-pad kernel.  Quacks like the (B,T,C,Hp,Wp) tensor where the
    trainer only needs the shape."""

    def __init__(self, ds: "SyntheticE33OMA_CRNN", t0: np.ndarray):
        self.ds, self.t0 = ds, np.asarray(t0, dtype=np.int32)
        Hp, Wp = ds.padding if ds.padding else ds.grid
        self.shape = (len(self.t0), ds.seq_len, ds.in_channels, Hp, Wp)
        self.device = ds.device

    def fill_slab(self, eng, ws):
        ds = self.ds
        dv = ds._device_arrays()
        B, T, Cc, Hp, Wp = self.shape
        assert (ws.B, ws.T, ws.H, ws.W) == (B, T, Hp, Wp) and Cc == eng.cfgs[0].Cx
        ptrs, lev = ds._sources(dv)
        H, W = ds.grid
        t0 = (C.c_int * B)(*[int(t) for t in self.t0])
        check(_lib.load().nint_preproc_fuse_pad_slab(ptrs, lev, len(lev), ptr(dv["mean"]), ptr(dv["std"]), t0, B,
                                                     ptr(ws.xs), ws.Cxp0, eng.cfgs[0].k if eng.cfgs[0].xfold else 0, T, H, W,
                                                     C.byref(ws.g), ds.mode, eng.dt,
                                                     stream_ptr()), "nint_preproc_fuse_pad_slab")


class SyntheticE33OMA_CRNN(torch.utils.data.Dataset):
    def __init__(self, period: str, species: str = "bcb", padding: Tuple[int, int] = (100, 154), in_channels: int = 5,
                 sequence_length: int = 10, *, levels: int = 1, n_steps: int = 480, grid: Tuple[int, int] = (90, 144),
                 pad_mode: str = "reference", device="cuda", seed: int = 0):
        super().__init__()
        assert species == "bcb", "only the BCB statistics ship with the reference"
        self.period, self.padding, self.seq_len, self.levels = period, tuple(padding) if padding else None, sequence_length, levels
        self.in_channels, self.grid, self.device = in_channels, tuple(grid), torch.device(device)
        self.mode = {"reference": 0, "reflect": 1}[pad_mode]
        self.generic = in_channels != 3 * levels + 2       # static attributes (dataset.py:100-122) are out of scope
        H, W = self.grid
        rng = np.random.default_rng(seed)

        def field(name, shape, positive=False):
            m, s = STATS[name]
            a = rng.standard_normal(shape).astype(np.float32)
            if positive:   # precipitation / emission / concentration: non-negative, heavy right tail
                a = np.maximum(0.0, m + s * (np.exp(0.9 * a) - 1.2)).astype(np.float32)
            else:
                a = (m + s * a).astype(np.float32)
            return a
        if self.generic:
            # `in_channels` generic fields with the wind statistics; one source of in_channels "levels"
            self.gen = field("u", (n_steps, in_channels, H, W))
            self.fields = [("gen", self.gen)]
        else:
            self.u = field("u", (n_steps, levels, H, W))
            self.v = field("v", (n_steps, levels, H, W))
            self.w = field("w", (n_steps, levels, H, W))
            self.prec = field("prec", (n_steps, H, W), True)
            self.src = field("bc_src", (n_steps, H, W), True)
            self.fields = [("u", self.u), ("v", self.v), ("w", self.w), ("prec", self.prec), ("src", self.src)]
        self.yraw = field("bc_conc", (n_steps, levels, H, W), True)
        ntrain = int(round(0.7 * n_steps))      # the reference hard-codes 3023 of 4320 steps (70 %)
        nval = int(round(0.1 * n_steps))
        # statistics over the training part of the record (dataset.py:589-596), one per fused channel
        chans = []
        for _, a in self.fields:
            chans += [a[:ntrain, l] for l in range(a.shape[1])] if a.ndim == 4 else [a[:ntrain]]
        self.X_mean = np.array([c.mean() for c in chans], dtype=np.float32)
        self.X_std = np.array([c.std() for c in chans], dtype=np.float32)
        self.y_mean = np.float32(self.yraw[:ntrain].mean())
        self.y_std = np.float32(self.yraw[:ntrain].std())
        nseq = n_steps - sequence_length + 1
        lo, hi = {"train": (0, ntrain), "val": (ntrain, ntrain + nval), "test": (ntrain + nval, nseq)}[period]
        self.first = np.arange(lo, min(hi, nseq))        # first time index of each window
        self._dev = None

    def __len__(self):
        return len(self.first)

    # ---- host-side pieces (testable without a GPU)
    def window(self, index: int):
        """raw (un-normalised, un-padded) window and target of sample `index` (dataset.py:614-616,599)."""
        t0 = int(self.first[index])
        sl = slice(t0, t0 + self.seq_len)
        return tuple(a[sl] for _, a in self.fields), self.yraw[t0 + self.seq_len - 1]

    # ---- device-side pieces
    def _device_arrays(self):
        if self._dev is None:
            d = self.device
            self._dev = {name: torch.from_numpy(a).to(d) for name, a in self.fields}
            self._dev.update(y=torch.from_numpy(self.yraw).to(d), mean=torch.from_numpy(self.X_mean).to(d),
                             std=torch.from_numpy(self.X_std).to(d),
                             ymean=torch.full((self.levels,), float(self.y_mean), device=d),
                             ystd=torch.full((self.levels,), float(self.y_std), device=d))
        return self._dev

    def _sources(self, dv):
        """(host array of record base pointers, host array of levels per source) in fusion order (dataset.py:526)"""
        n = len(self.fields)
        ptrs = (C.c_void_p * n)(*[dv[name].data_ptr() for name, _ in self.fields])
        lev = (C.c_int * n)(*[a.shape[1] if a.ndim == 4 else 1 for _, a in self.fields])
        return ptrs, lev

    def _targets(self, dv, t0s):
        """z-scored targets (dataset.py:596,599) of a batch: the tracer at each window's LAST step, one launch"""
        H, W = self.grid
        B, L = len(t0s), self.levels
        y = torch.empty(B, L, H, W, dtype=torch.float32, device=self.device)
        yptr = (C.c_void_p * 1)(dv["y"].data_ptr())
        ylev = (C.c_int * 1)(L)
        tl = (C.c_int * B)(*[int(t) + self.seq_len - 1 for t in t0s])
        check(_lib.load().nint_preproc_fuse_pad_batch(yptr, ylev, 1, ptr(dv["ymean"]), ptr(dv["ystd"]), tl, B, ptr(y),
                                                      1, H, W, H, W, 1, stream_ptr()), "target z-score")
        return y[:, 0] if L == 1 else y

    def device_batch(self, indices: Sequence[int]):
        """(X (B,T,C,Hp,Wp) f32, y (B,[L,]H,W) f32) on the GPU: the reference's tensors (dataset.py:538-539),
        one launch of the fuse/z-score/halo-pad kernel for the whole batch."""
        dv = self._device_arrays()
        H, W = self.grid
        Hp, Wp = self.padding if self.padding else (H, W)
        t0s = [int(self.first[int(i)]) for i in indices]
        B, T = len(t0s), self.seq_len
        X = torch.empty(B, T, self.in_channels, Hp, Wp, dtype=torch.float32, device=self.device)
        ptrs, lev = self._sources(dv)
        t0 = (C.c_int * B)(*t0s)
        check(_lib.load().nint_preproc_fuse_pad_batch(ptrs, lev, len(lev), ptr(dv["mean"]), ptr(dv["std"]), t0, B, ptr(X),
                                                      T, H, W, Hp, Wp, self.mode, stream_ptr()), "nint_preproc_fuse_pad_batch")
        return X, self._targets(dv, t0s)

    def slab_batch(self, indices: Sequence[int]):
        """(SlabBatch, y): the same batch, X left un-materialised -- the engine writes it straight into its input
        slab (bf16 or f32, channels-last) when the trainer / model consumes it."""
        dv = self._device_arrays()
        t0s = [int(self.first[int(i)]) for i in indices]
        return SlabBatch(self, np.asarray(t0s)), self._targets(dv, t0s)

    def __getitem__(self, index):
        X, y = self.device_batch([index])
        return X[0], y[0]
