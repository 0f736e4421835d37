"""The 8-wave LDS-weight gate kernel (csrc/conv_wide.hip, nint_layer.wide) against the 4-wave kernel and the oracle.

Both kernels run the same K order (chunk-major, taps row by row) into f32 accumulators that start at the bias, and the
4-wave kernel does not slice K when the gate columns are a multiple of 256 -- so the two must agree BIT FOR BIT on h, c and
the gate stash, whatever the tile shape (R rows x 16/R blocks, leftover strips, tiles that overhang the slab's slack),
the number of tiles per persistent workgroup, the chunk-ring depth or the number of column groups.  The oracle comparison
of the forced-wide path at the bench's size closes the loop (tolerances: the suite's bf16 ones, rel-L2 <= 2e-2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as p
    p.load_library()
    return p


def _forward_slabs(Cx, hidden, ks, B, T, H, W, wide, seed):
    from nasa_niswan_amd import engine
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    engine.FORCE_WIDE = wide
    # the 4-wave reference with its launch shape pinned (8-row tiles, all gate columns in one workgroup: no K slices): small
    # batches would otherwise split the columns over workgroups whose waves slice K -- another summation order
    engine.FORCE_TILE_ROWS = 8          # (for every engine: layers the wide kernel does not serve fall back to the 4-wave one)
    try:
        eng = SeqEngine([LayerCfg(Cx if l == 0 else hidden[l - 1], hidden[l], ks[l]) for l in range(len(hidden))], "bf16", "cuda")
    finally:
        engine.FORCE_WIDE = 0
        engine.FORCE_TILE_ROWS = 0
    g = torch.Generator(device="cuda").manual_seed(seed)
    Ws = [torch.randn(4 * c.Ch, c.Cx + c.Ch, c.k, c.k, device="cuda", generator=g) * 0.05 for c in eng.cfgs]
    bs = [torch.randn(4 * c.Ch, device="cuda", generator=g) * 0.2 for c in eng.cfgs]
    X = torch.randn(B, T, Cx, H, W, device="cuda", generator=g)
    eng.pack_weights(Ws, bs)
    ws = eng.acquire(B, T, H, W, True, False)
    eng.forward(ws, X)
    torch.cuda.synchronize()
    out = [(ws.h[l].clone(), ws.c[l].clone(), ws.gates[l].clone()) for l in range(len(hidden))]
    eng.release(ws)
    return out


SHAPES = {
    # name: (Cx, hidden, ks, B, T, H, W)
    "bench-layer0-B2": (62, [64], [5], 2, 3, 100, 154),                # 16x16 tiles + a 4x64 strip that overhangs the slab's slack
    "bench-layer0-B8-two-tiles-per-workgroup": (62, [64], [5], 8, 2, 100, 154),
    "tiny-grid-one-tile": (6, [64], [3], 1, 2, 9, 17),
    "two-column-groups-k3": (70, [128], [3], 2, 2, 37, 50),
    "odd-grid-k5-three-images": (40, [64], [5], 3, 2, 23, 41),
    "stack-64-64": (5, [64, 64], [3, 5], 2, 2, 30, 45),                # second layer: x source = the first layer's h slab
    "wide-strip-only": (8, [64], [3], 2, 2, 3, 200),                   # H < the smallest tile height that pays: flat tiles
    "tall-narrow": (8, [64], [5], 2, 2, 130, 11),
    "cfg3-layer-hidden128-190x298": (62, [128], [3], 1, 2, 190, 298),   # BASELINE configs[3] geometry: 4 / 2 column-group sets per tile
}


@pytest.mark.parametrize("wide", [3 + 8, 4 + 8], ids=["256px-tiles", "512px-tiles"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_wide_kernel_is_bit_identical_to_the_4wave_kernel(pkg, name, wide):
    """wide = 3 / 4 (+ 8: taps in plain order in every workgroup)."""
    a = _forward_slabs(*SHAPES[name], wide=1, seed=3)
    b = _forward_slabs(*SHAPES[name], wide=wide, seed=3)
    for l, (x, y) in enumerate(zip(a, b)):
        for what, u, v in zip(("h", "c", "gates"), x, y):
            assert bool(torch.isfinite(v.view(torch.bfloat16 if what != "c" else torch.float32).float()).all()), (name, l, what)
            same = torch.equal(u, v)
            if not same:
                d = (u.view(torch.uint8) != v.view(torch.uint8)).float().mean()
                raise AssertionError(f"{name}: layer {l} {what} differs in {100 * float(d):.3f} % of its bytes")


@pytest.mark.parametrize("wide", [3, 4], ids=["256px-tiles", "512px-tiles"])
@pytest.mark.parametrize("name", ["bench-layer0-B8-two-tiles-per-workgroup", "two-column-groups-k3", "odd-grid-k5-three-images", "stack-64-64"])
def test_rotated_tap_order_differs_only_in_the_last_bits(pkg, name, wide):
    """The product order: every workgroup starts a chunk's taps at its own tap (L2 hot-spot avoidance).  Same products, another
    f32 summation order: h / gates (bf16) and c (f32) agree with the 4-wave kernel to rounding, two time steps deep."""
    a = _forward_slabs(*SHAPES[name], wide=1, seed=3)
    b = _forward_slabs(*SHAPES[name], wide=wide, seed=3)
    for l, (x, y) in enumerate(zip(a, b)):
        for what, u, v in zip(("h", "c", "gates"), x, y):
            dt = torch.float32 if what == "c" else torch.bfloat16
            uf, vf = u.view(dt).double(), v.view(dt).double()
            assert bool(torch.isfinite(vf).all()), (name, l, what)
            r = float((uf - vf).norm() / (uf.norm() + 1e-30))
            assert r <= (2e-5 if what == "c" else 2e-3), (name, l, what, r)


@pytest.mark.parametrize("wide", [1, 3, 4])
def test_forced_kernel_family_trains_the_bench_workload_like_the_oracle(pkg, wide):
    """cfg1-20level at B=2 (train step, bf16) with the kernel family forced: prediction, loss and all 8 gradients."""
    from nasa_niswan_amd import engine
    from test_gpu_fullsize import CFG1, _check, _fit_step_both
    engine.FORCE_WIDE = wide
    try:
        _check(_fit_step_both(pkg, B=2, dtype="bf16", **CFG1), "bf16")
    finally:
        engine.FORCE_WIDE = 0


def test_every_family_code_runs_every_batch(pkg):
    """nint_layer.wide = 0 (the library's choice: the 4-wave kernel, which measured faster -- DESIGN.md 4.4), 2 (the wide
    kernel wherever instantiated), 3 / 4 (its tile size forced), all with the plain tap order (+ 8): bit-identical results
    at B = 1 (63 tiles: most CUs idle), 2 and 8 (two units per persistent workgroup)."""
    for B in (1, 2, 8):
        ref = _forward_slabs(62, [64], [5], B, 2, 100, 154, wide=1, seed=5)
        for w in (0, 2 + 8, 3 + 8, 4 + 8):
            got = _forward_slabs(62, [64], [5], B, 2, 100, 154, wide=w, seed=5)
            assert all(torch.equal(u, v) for x, y in zip(ref, got) for u, v in zip(x, y)), (B, w)
