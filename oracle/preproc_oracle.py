"""CPU (numpy) restatement of the reference pre-processing tail: channel fusion,
z-score, cyclic-longitude / "reflective"-latitude halo pad.  TEST INFRASTRUCTURE ONLY.

Follows reference dataset.py:21-58 (3-D base class), :67-98 (4-D RNN override),
:520-536 and :587-599 (stack + z-score).  The 3-D pad is pinned by the notebook's
13x13 matrix (dataset_config.ipynb:484-502).  The 4-D RNN pad has NO golden in the
reference; it reproduces the committed code literally, including the ``np.fliplr``
quirk (dataset.py:96 flips axis 1 = channels, not latitude).
"""
from __future__ import annotations

import numpy as np

__all__ = ["cyclic_pad", "lat_pad_3d", "lat_pad_4d_quirk", "padding_data_3d", "padding_data_4d",
           "fuse_levels", "zscore", "preproc_sample", "NOTEBOOK_13x13"]

# dataset_config.ipynb:484-502 (cell 15 output): padding_data(arange(25).reshape(1,5,5), 13)
NOTEBOOK_13x13 = np.array([
    [21, 22, 23, 24, 20, 21, 22, 23, 24, 20, 21, 22, 23],
    [16, 17, 18, 19, 15, 16, 17, 18, 19, 15, 16, 17, 18],
    [11, 12, 13, 14, 10, 11, 12, 13, 14, 10, 11, 12, 13],
    [6, 7, 8, 9, 5, 6, 7, 8, 9, 5, 6, 7, 8],
    [1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1, 2, 3],
    [6, 7, 8, 9, 5, 6, 7, 8, 9, 5, 6, 7, 8],
    [11, 12, 13, 14, 10, 11, 12, 13, 14, 10, 11, 12, 13],
    [16, 17, 18, 19, 15, 16, 17, 18, 19, 15, 16, 17, 18],
    [21, 22, 23, 24, 20, 21, 22, 23, 24, 20, 21, 22, 23],
    [16, 17, 18, 19, 15, 16, 17, 18, 19, 15, 16, 17, 18],
    [11, 12, 13, 14, 10, 11, 12, 13, 14, 10, 11, 12, 13],
    [6, 7, 8, 9, 5, 6, 7, 8, 9, 5, 6, 7, 8],
    [1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1, 2, 3]], dtype=np.int64)[None]


def cyclic_pad(data: np.ndarray, wp: int) -> np.ndarray:
    """dataset.py:21-34 (3-D) / :67-80 (4-D): cyclic extension along the last axis."""
    w = data.shape[-1]
    pad_left = (wp - w) // 2
    pad_right = wp - w - pad_left
    if pad_left <= w and pad_right <= w:
        return np.concatenate([data[..., -pad_left:], data, data[..., :pad_right]], axis=data.ndim - 1)
    raise AttributeError("The requested padding size is larger than width size of the input image.")


def lat_pad_3d(data: np.ndarray, hp: int) -> np.ndarray:
    """dataset.py:36-53: (C,H,W) arrays; ``np.fliplr`` flips axis 1 = latitude, so this is a
    true reflect (edge row excluded)."""
    assert data.ndim == 3
    h = data.shape[1]
    pad_top = (hp - h) // 2
    pad_bottom = hp - h - pad_top
    pad_top += 1
    pad_bottom += 1
    if pad_top <= h and pad_bottom <= h:
        return np.concatenate((np.fliplr(data[:, 1:pad_top]), data, np.fliplr(data[:, -pad_bottom:-1])), axis=1)
    raise AttributeError("The requested padding size is larger than height size of the input image.")


def lat_pad_4d_quirk(data: np.ndarray, hp: int) -> np.ndarray:
    """dataset.py:82-98: (T,C,H,W) arrays; ``np.fliplr`` still flips axis 1, which is now
    the CHANNEL axis: halo rows keep their latitude order and come from channel C-1-c."""
    assert data.ndim == 4
    h = data.shape[2]
    pad_top = (hp - h) // 2
    pad_bottom = hp - h - pad_top
    pad_top += 1
    pad_bottom += 1
    if pad_top <= h and pad_bottom <= h:
        return np.concatenate((np.fliplr(data[:, :, 1:pad_top]), data,
                               np.fliplr(data[:, :, -pad_bottom:-1])), axis=2)
    raise AttributeError("The requested padding size is larger than height size of the input image.")


def padding_data_3d(data: np.ndarray, padding) -> np.ndarray:
    """dataset.py:55-58 with (Hp, Wp) = padding."""
    return lat_pad_3d(cyclic_pad(data, padding[1]), padding[0])


def padding_data_4d(data: np.ndarray, padding, mode: str = "reference") -> np.ndarray:
    """dataset.py:55-58 through the RNN overrides (:67-98).  ``mode='reflect'`` applies the
    3-D semantics per time step instead (what the authors evidently intended)."""
    data = cyclic_pad(data, padding[1])
    if mode == "reference":
        return lat_pad_4d_quirk(data, padding[0])
    if mode == "reflect":
        return np.stack([lat_pad_3d(d, padding[0]) for d in data], axis=0)
    raise ValueError(mode)


def fuse_levels(u, v, w, prec, src) -> np.ndarray:
    """dataset.py:456-460,513,526: ``np.stack([u, v, omega, prec, src], axis=1)`` with
    u,v,omega taken at level 0.  Extension defined by this build (no reference code;
    SURVEY.md section 8 a-6): u,v,omega given as (T,L,H,W) are laid out as 3*L level-channels
    followed by the two 2-D fields -> (T, 3L+2, H, W).  L=1 is the reference."""
    u, v, w = (a[:, None] if a.ndim == 3 else a for a in (u, v, w))
    return np.concatenate([u, v, w, prec[:, None], src[:, None]], axis=1)


def zscore(x: np.ndarray, mean, std) -> np.ndarray:
    """dataset.py:520-521,528: per-channel float32 mean/std reshaped (1,C,1,1)."""
    mean = np.asarray(mean, dtype=np.float32).reshape(1, -1, 1, 1)
    std = np.asarray(std, dtype=np.float32).reshape(1, -1, 1, 1)
    return (x - mean) / std


def preproc_sample(u, v, w, prec, src, mean, std, padding=None, mode: str = "reference") -> np.ndarray:
    """dataset.py:526-539: fuse -> z-score -> pad -> float32."""
    x = zscore(fuse_levels(u, v, w, prec, src), mean, std)
    if padding:
        x = padding_data_4d(x, padding, mode)
    return x.astype(np.float32)


def inmemory_rnn_dataset(X1, X2, X3, X4, X5, y, period: str, seq_len: int):
    """dataset.py:584-616 (`E33OMA90D_CRNN._get_data`) on arrays instead of the NetCDF file: stack, statistics over the first
    3023 steps, z-score, sliding windows, target lag, 3023 / 3455 split.  Returns (X (n, T, 5, H, W), y (n, H, W), X_mean, X_std,
    y_mean, y_std) exactly as the reference's attributes hold them (before padding)."""
    Xs = np.stack([X1, X2, X3, X4, X5], axis=1)                             # dataset.py:584
    y_mean = y[:3023, ...].mean().reshape(-1, 1, 1)                         # :587
    y_std = y[:3023, ...].std().reshape(-1, 1, 1)
    X_mean = Xs[:3023, ...].mean(axis=(0, 2, 3)).reshape(-1, 1, 1)          # :590
    X_std = Xs[:3023, ...].std(axis=(0, 2, 3)).reshape(-1, 1, 1)
    Xs = (Xs - X_mean) / X_std                                              # :593
    y = (y - y_mean) / y_std
    X = np.lib.stride_tricks.sliding_window_view(Xs, (seq_len, *Xs.shape[1:])).squeeze()     # :614-616
    y = y[seq_len - 1:]                                                     # :599
    sl = {"train": slice(None, 3023), "val": slice(3023, 3455), "test": slice(3455, None)}[period]   # :601-612
    return X[sl], y[sl], X_mean, X_std, y_mean, y_std
