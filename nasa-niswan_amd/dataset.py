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
levels; L=1 is the reference."""
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


class SyntheticE33OMA_CRNN(torch.utils.data.Dataset):
    def __init__(self, period: str, species: str = "bcb", padding: Tuple[int, int] = (100, 154), in_channels: int = 5,
                 sequence_length: int = 10, *, levels: int = 1, n_steps: int = 480, grid: Tuple[int, int] = (90, 144),
                 pad_mode: str = "reference", device="cuda", seed: int = 0):
        super().__init__()
        assert species == "bcb", "only the BCB statistics ship with the reference"
        assert in_channels == 3 * levels + 2, "in_channels must equal 3*levels+2 (static attributes are out of scope)"
        self.period, self.padding, self.seq_len, self.levels = period, tuple(padding) if padding else None, sequence_length, levels
        self.in_channels, self.grid, self.device = in_channels, grid, torch.device(device)
        self.mode = {"reference": 0, "reflect": 1}[pad_mode]
        H, W = grid
        rng = np.random.default_rng(seed)

        def field(name, shape, positive=False):
            m, s = STATS[name]
            a = rng.standard_normal(shape).astype(np.float32)
            if positive:   # precipitation / emission / concentration: non-negative, heavy right tail
                a = np.maximum(0.0, m + s * (np.exp(0.9 * a) - 1.2)).astype(np.float32)
            else:
                a = (m + s * a).astype(np.float32)
            return a
        self.u = field("u", (n_steps, levels, H, W))
        self.v = field("v", (n_steps, levels, H, W))
        self.w = field("w", (n_steps, levels, H, W))
        self.prec = field("prec", (n_steps, H, W), True)
        self.src = field("bc_src", (n_steps, H, W), True)
        self.yraw = field("bc_conc", (n_steps, levels, H, W), True)
        ntrain = int(round(0.7 * n_steps))      # the reference hard-codes 3023 of 4320 steps (70 %)
        nval = int(round(0.1 * n_steps))
        # statistics over the training part of the record (dataset.py:589-596)
        chans = [self.u[:ntrain, l] for l in range(levels)] + [self.v[:ntrain, l] for l in range(levels)] + \
                [self.w[:ntrain, l] for l in range(levels)] + [self.prec[:ntrain], self.src[:ntrain]]
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
        return (self.u[sl], self.v[sl], self.w[sl], self.prec[sl], self.src[sl]), self.yraw[t0 + self.seq_len - 1]

    # ---- device-side pieces
    def _device_arrays(self):
        if self._dev is None:
            d = self.device
            self._dev = dict(u=torch.from_numpy(self.u).to(d), v=torch.from_numpy(self.v).to(d), w=torch.from_numpy(self.w).to(d),
                             prec=torch.from_numpy(self.prec).to(d), src=torch.from_numpy(self.src).to(d),
                             y=torch.from_numpy(self.yraw).to(d), mean=torch.from_numpy(self.X_mean).to(d),
                             std=torch.from_numpy(self.X_std).to(d))
        return self._dev

    def device_batch(self, indices: Sequence[int]):
        """(X (B,T,C,Hp,Wp) f32, y (B,[L,]H,W) f32) on the GPU: the whole record stays resident in HBM and
        every sample is one launch of the fuse/z-score/halo-pad kernel on its window (pointer offsets)."""
        dv = self._device_arrays()
        lib = _lib.load()
        H, W = self.grid
        Hp, Wp = self.padding if self.padding else (H, W)
        B, T, L = len(indices), self.seq_len, self.levels
        X = torch.empty(B, T, self.in_channels, Hp, Wp, dtype=torch.float32, device=self.device)
        y = torch.empty(B, L, H, W, dtype=torch.float32, device=self.device)
        if "ymean" not in dv:
            dv["ymean"] = torch.full((L,), float(self.y_mean), device=self.device)
            dv["ystd"] = torch.full((L,), float(self.y_std), device=self.device)
        lev = (C.c_int * 5)(L, L, L, 1, 1)
        ylev = (C.c_int * 1)(L)
        st = stream_ptr()
        for b, idx in enumerate(indices):
            t0 = int(self.first[int(idx)])
            ptrs = (C.c_void_p * 5)(dv["u"][t0].data_ptr(), dv["v"][t0].data_ptr(), dv["w"][t0].data_ptr(),
                                    dv["prec"][t0].data_ptr(), dv["src"][t0].data_ptr())
            check(lib.nint_preproc_fuse_pad(ptrs, lev, 5, ptr(dv["mean"]), ptr(dv["std"]), ptr(X[b]), T, H, W, Hp, Wp,
                                            self.mode, st), "nint_preproc_fuse_pad")
            # target z-score (dataset.py:596) through the same kernel: one source, no halo
            yptr = (C.c_void_p * 1)(dv["y"][t0 + T - 1].data_ptr())
            check(lib.nint_preproc_fuse_pad(yptr, ylev, 1, ptr(dv["ymean"]), ptr(dv["ystd"]), ptr(y[b]), 1, H, W, H, W,
                                            1, st), "target z-score")
        if L == 1:
            y = y[:, 0]
        return X, y

    def __getitem__(self, index):
        X, y = self.device_batch([index])
        return X[0], y[0]
