"""The tiny-layer paths of the gate step: the dense-K MFMA kernel (csrc/tiny_gemm.hip, the library's choice) and the
vector-ALU stencil kernel (csrc/stencil.hip) -- SURVEY.md section 7 step 4 / north_star: "MFMA used only
... when channel count makes it a real dense contraction", reference op model.py:207-231 with ConvLSTM(4, [8], [3], 1)
(BASELINE configs[0]).  nint_cell_fwd takes it for nint_layer.tile_rows == 1 ("one pixel per lane") on layers with Ch <= 8,
k = 3 and thin inputs; left to itself (tile_rows == 0) the library runs such layers on the matrix pipe with a DENSE K (16-byte
channel groups of the staged halo tile gathered by the four lane groups of one fragment read: 4 K-steps instead of 12 for
configs[0]; its choice in f32 storage) -- so every shape here runs THREE families: the dense-K kernel (engine.FORCE_TILE_ROWS = 2), the stencil kernel
(= 1) and the padded implicit-GEMM kernel (= 8), each against the CPU oracle (prediction, loss-weighted gradients of every parameter and of the
input) and against each other.

Tolerances: f32 outputs rtol 1e-4 / atol 1e-5 and gradients max-abs <= 1e-3 max|g| against the oracle, the two families within
1e-5 of the output's max (another f32 summation order); bf16 rel-L2 <= 2e-2 against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (C, hidden, B, T, H, W): configs[0] itself; a ragged grid with 5 real hidden channels (a half-empty channel quad) and a
# first layer that is too wide to fold; stacked tiny layers (layer 1 reads the 8-channel h of layer 0 UNFOLDED: horizontal taps
# by DPP row shifts, edge lanes from the halo column); 16 folded channels (48 folded: 12 quads); one hidden channel.
SHAPES = {
    "cfg0": (4, [8], 2, 4, 32, 32),
    "ragged-5-hidden": (3, [5], 3, 2, 13, 37),
    "stack-8-8-4": (4, [8, 8, 4], 2, 3, 19, 50),
    "16-in-folded": (16, [8], 1, 2, 9, 33),
    "one-hidden-channel": (2, [1, 8], 2, 2, 8, 16),
}


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as p
    p.load_library()
    return p


def _run(pkg, params, C_, hidden, X, wgt, dtype, rows):
    from nasa_niswan_amd import engine
    engine.FORCE_TILE_ROWS = rows
    try:
        net = pkg.ConvLSTM(C_, hidden, [3] * len(hidden), len(hidden), out_channels=wgt.shape[1], compute_dtype=dtype).cuda()
        net.load_state_dict(params)
        Xd = X.cuda().requires_grad_(True)
        pred = net(Xd)
        (pred * wgt.cuda()).sum().backward()
        torch.cuda.synchronize()
        eng = net._engine(Xd.device)
        held = [bool(pkg.load_library().nint_stencil_holds(C.byref(ly))) for ly in eng.layers]
    finally:
        engine.FORCE_TILE_ROWS = 0
    res = {"pred": pred.detach().cpu(), "dX": Xd.grad.cpu()}
    for k, p in net.named_parameters():
        res["grad." + k] = p.grad.cpu()
    return res, held


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_stencil_and_gemm_families_against_the_oracle_and_each_other(pkg, name, dtype):
    from oracle import convlstm_oracle as O
    C_, hidden, B, T, H, W = SHAPES[name]
    L, out = len(hidden), 2
    rng = np.random.default_rng(77)
    params = O.synth_params(C_, hidden, [3] * L, L, out_channels=out, seed=9)
    X = torch.from_numpy(rng.standard_normal((B, T, C_, H, W)).astype(np.float32))
    wgt = torch.from_numpy(rng.standard_normal((B, out, H, W)).astype(np.float32))
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Xo = X.clone().requires_grad_(True)
    po = O.convlstm_forward(Xo, leaf)
    (po * wgt).sum().backward()
    ref = {"pred": po.detach(), "dX": Xo.grad}
    for k in params:
        ref["grad." + k] = leaf[k].grad
    st, held = _run(pkg, params, C_, hidden, X, wgt, dtype, 1)
    mm, _ = _run(pkg, params, C_, hidden, X, wgt, dtype, 8)
    dk, _ = _run(pkg, params, C_, hidden, X, wgt, dtype, 2)
    assert all(held), (name, held)                               # every layer of these stacks is a stencil shape
    for fam, res in (("dense-K", dk), ("stencil", st), ("gemm", mm)):
        for k, a in res.items():
            a, b = a.double(), ref[k].double()
            assert torch.isfinite(a).all(), (fam, k)
            if dtype == "f32":
                if k == "pred":
                    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), (fam, k, float((a - b).abs().max()))
                else:
                    assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-9, (fam, k)
            else:
                e = float((a - b).norm() / (b.norm() + 1e-30))
                assert e <= 2e-2, (fam, k, e)
    d = float((st["pred"].double() - mm["pred"].double()).abs().max() / mm["pred"].double().abs().max())
    d2 = float((dk["pred"].double() - mm["pred"].double()).abs().max() / mm["pred"].double().abs().max())
    print(f"  {name} {dtype}: prediction, max diff / max: stencil vs gemm {d:.2e}, dense-K vs gemm {d2:.2e}")
    assert d <= (1e-5 if dtype == "f32" else 2e-2) and d2 <= (1e-5 if dtype == "f32" else 2e-2), (name, d, d2)


def test_stencil_choice_is_host_arithmetic(pkg):
    from nasa_niswan_amd._lib import NintLayer
    lib = pkg.load_library()

    def holds(Cx, Ch, k, xfold=0):
        ly = NintLayer()
        ly.Cx, ly.Ch, ly.k, ly.xfold = Cx, Ch, k, xfold
        return bool(lib.nint_stencil_holds(C.byref(ly)))

    assert holds(4, 8, 3, 1) and holds(8, 8, 3, 0) and holds(16, 4, 3, 0) and holds(21, 8, 3, 1)
    assert not holds(4, 16, 3, 1) and not holds(4, 8, 5, 1) and not holds(17, 8, 3, 0) and not holds(22, 8, 3, 1)
