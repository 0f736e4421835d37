"""Inference helpers around the HIP forward path, mirroring the reference's analysis notebook:
test-set prediction with de-normalisation (test.ipynb cell 8, :257-300) and the one-at-a-time
(OAT) input-perturbation sweep (test.ipynb cell 56, :2433-2461).  Forward only, `torch.no_grad()`."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch


@torch.no_grad()
def predict(net, dataset, batch_size: int = 8, halo: Tuple[int, int] = (5, 5), indices: Sequence[int] = None):
    """Returns (GTs, PDs[, HSs]) in physical units: crop `[halo:halo+H]`, squeeze, `p * y_std + y_mean`
    (test.ipynb cell 8).  HSs (per-step head outputs) only when `net.return_sequence` is set."""
    net.eval()
    idx = list(range(len(dataset))) if indices is None else list(indices)
    H, W = dataset.grid
    gts, pds, hss = [], [], []
    for s in range(0, len(idx), batch_size):
        X, y = dataset.device_batch(idx[s:s + batch_size])
        out = net(X)
        pred, hs = (out if isinstance(out, tuple) else (out, None))
        p = pred[:, :, halo[0]:halo[0] + H, halo[1]:halo[1] + W]
        gts.append(y.cpu().numpy().reshape(p.shape) * dataset.y_std + dataset.y_mean)
        pds.append(p.cpu().numpy() * dataset.y_std + dataset.y_mean)
        if hs is not None:
            hss.append(hs[:, :, halo[0]:halo[0] + H, halo[1]:halo[1] + W].cpu().numpy() * dataset.y_std + dataset.y_mean)
    res = (np.concatenate(gts), np.concatenate(pds))
    return res + (np.concatenate(hss),) if hss else res


@torch.no_grad()
def oat_sensitivity(net, dataset, num_ftrs: int = 5, perturbed_values: float = 0.05, batch_size: int = 8,
                    halo: Tuple[int, int] = (5, 5), indices: Sequence[int] = None) -> np.ndarray:
    """One-at-a-time sweep (test.ipynb cell 56): for feature i, `X[:, :, i] *= 1 + perturbed_values` on the
    normalised, padded input, forward, crop, de-normalise.  Returns (num_ftrs, N, O, H, W)."""
    net.eval()
    idx = list(range(len(dataset))) if indices is None else list(indices)
    H, W = dataset.grid
    outs = []
    for i in range(num_ftrs):
        pds = []
        for s in range(0, len(idx), batch_size):
            X, _ = dataset.device_batch(idx[s:s + batch_size])
            X[:, :, i] *= (1 + perturbed_values)
            out = net(X)
            pred = out[0] if isinstance(out, tuple) else out
            p = pred[:, :, halo[0]:halo[0] + H, halo[1]:halo[1] + W]
            pds.append(p.cpu().numpy() * dataset.y_std + dataset.y_mean)
        outs.append(np.concatenate(pds))
    return np.stack(outs)
