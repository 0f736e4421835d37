#!/usr/bin/env python3
"""Do two launch schedules (nint_seq.wave values) give the same bits?  One train step of the bench recipe at a small T, every
output compared byte for byte.   python tools/wave_bits.py 2 18 [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nasa_niswan_amd as pkg  # noqa: E402
from nasa_niswan_amd import engine as _engine  # noqa: E402


def run(wave, B, T, dtype):
    _engine.FORCE_WAVE = wave
    torch.manual_seed(0)
    m = pkg.ConvLSTM(62, [64, 32, 16], [5, 3, 3], 3, out_channels=20, compute_dtype=dtype).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, T, 62, 100, 154, device="cuda", generator=g)
    x.requires_grad_(True)
    y = m(x)
    y.square().mean().backward()
    torch.cuda.synchronize()
    return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("waves", type=int, nargs=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    a = run(args.waves[0], args.batch, args.T, args.dtype)
    b = run(args.waves[1], args.batch, args.T, args.dtype)
    worst = 0.0
    for i, (u, v) in enumerate(zip(a, b)):
        same = torch.equal(u, v)
        rel = float((u - v).norm() / (u.norm() + 1e-30))
        worst = max(worst, rel)
        print(f"tensor {i} {tuple(u.shape)}: {'same bits' if same else f'rel-L2 {rel:.3e}'}")
    print(f"wave {args.waves[0]} vs {args.waves[1]}: worst rel-L2 {worst:.3e}")


if __name__ == "__main__":
    main()
