#!/usr/bin/env python3
"""Digest of one training step's results (loss, every gradient, the weights after Adam) on the bench workload, for comparing
two builds of the library bit for bit:  python tools/ab_grads.py [--lib other.so] [--batch 2] [--dtype bf16]"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--hidden", default="64,32,16")
    ap.add_argument("--kernels", default="5,3,3")
    args = ap.parse_args()
    import nasa_niswan_amd as pkg
    pkg.load_library(args.lib) if args.lib else pkg.load_library()
    from nasa_niswan_amd.trainer import FusedTrainer
    hidden = [int(v) for v in args.hidden.split(",")]
    ks = [int(v) for v in args.kernels.split(",")]
    torch.manual_seed(0)
    net = pkg.ConvLSTM(62, hidden, ks, len(hidden), out_channels=20, compute_dtype=args.dtype).cuda()
    tr = FusedTrainer(net, lr=1e-3, halo=(5, 5))
    X = torch.randn(args.batch, 12, 62, 100, 154, device="cuda")
    y = torch.randn(args.batch, 20, 90, 144, device="cuda")
    loss = float(tr.step(X, y))
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for i, (k, p) in enumerate(net.named_parameters()):
        g = tr.flat.grad_view(i).detach().cpu().contiguous()
        h.update(g.numpy().tobytes())
        print(f"  {k:28s} grad sha {hashlib.sha256(g.numpy().tobytes()).hexdigest()[:12]}  |g| {float(g.norm()):.6e}")
    print(f"loss {loss:.9f}  digest {h.hexdigest()[:16]}")


if __name__ == "__main__":
    main()
