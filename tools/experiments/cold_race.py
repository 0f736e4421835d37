#!/usr/bin/env python3
"""Cold-process race screen: ONE forward of the bench's first layer (B=2, T=3) per kernel family in a fresh process -- the
4-wave kernel twice (its own determinism) and the 8-wave kernel (plain tap order) -- compared byte for byte.
    for i in $(seq 20); do python tools/cold_race.py || break; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nasa_niswan_amd as pkg  # noqa: E402
from nasa_niswan_amd import engine  # noqa: E402
from nasa_niswan_amd.engine import LayerCfg, SeqEngine  # noqa: E402


def run(wide, Wt, bs, X, B, T, H, W):
    engine.FORCE_WIDE, engine.FORCE_TILE_ROWS = wide, 8
    try:
        e = SeqEngine([LayerCfg(62, 64, 5)], "bf16", "cuda")
    finally:
        engine.FORCE_WIDE, engine.FORCE_TILE_ROWS = 0, 0
    e.pack_weights(Wt, bs)
    ws = e.acquire(B, T, H, W, True, False)
    e.forward(ws, X)
    torch.cuda.synchronize()
    return ws.h[0].clone(), ws.c[0].clone(), ws.gates[0].clone()


def main():
    pkg.load_library()
    order = sys.argv[1] if len(sys.argv) > 1 else "a"
    B, T, H, W = 2, 3, 100, 154
    g = torch.Generator(device="cuda").manual_seed(3)
    Wt = [torch.randn(256, 126, 5, 5, device="cuda", generator=g) * 0.05]
    bs = [torch.randn(256, device="cuda", generator=g) * 0.2]
    X = torch.randn(B, T, 62, H, W, device="cuda", generator=g)
    seq = (1, 11, 1) if order == "a" else (11, 1, 1)
    outs = [run(w, Wt, bs, X, B, T, H, W) for w in seq]
    ref = outs[seq.index(1)]
    bad = 0
    for w, o in zip(seq, outs):
        for name, a, b in zip(("h", "c", "gates"), o, ref):
            if not torch.equal(a, b):
                bad += 1
                d = (a.view(torch.uint8) != b.view(torch.uint8)).nonzero().flatten()
                print(f"run of wide={w} (order {seq}): {name} differs in {len(d)} bytes, first byte {int(d[0])}, last {int(d[-1])}")
    print("ok" if not bad else "MISMATCH", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
