"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/nint.h declares, and the ctypes structures match the header's layout.  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nint.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nint_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    import nasa_niswan_amd as pkg
    from nasa_niswan_amd import _lib
    lib = pkg.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in nint.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.nint_version() == _lib.NINT_VERSION == 111
    assert lib.nint_kc(0) == 16 and lib.nint_kc(1) == 32
    assert lib.nint_error_string(-2).decode().startswith("nint:")


def test_struct_layout_matches_header(tmp_path):
    """Compile a tiny C program against nint.h and compare sizeof/offsetof with ctypes."""
    from nasa_niswan_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nint.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(nint_geom),sizeof(nint_layer),sizeof(nint_seq),offsetof(nint_layer,Wf),offsetof(nint_seq,xs),'
                    'offsetof(nint_seq,dW),offsetof(nint_seq,wg_partial_bytes),offsetof(nint_layer,wide),'
                    'offsetof(nint_seq,probe),offsetof(nint_seq,probe_slots),offsetof(nint_seq,wave),sizeof(nint_seq));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.NintGeom), C.sizeof(_lib.NintLayer), C.sizeof(_lib.NintSeq), _lib.NintLayer.Wf.offset,
            _lib.NintSeq.xs.offset, _lib.NintSeq.dW.offset, _lib.NintSeq.wg_partial_bytes.offset, _lib.NintLayer.wide.offset,
            _lib.NintSeq.probe.offset, _lib.NintSeq.probe_slots.offset, _lib.NintSeq.wave.offset, C.sizeof(_lib.NintSeq)]
    assert got == want


def test_geometry_and_workspace_queries_are_host_only():
    from nasa_niswan_amd import _lib
    lib = _lib.load()
    g = _lib.NintGeom()
    assert lib.nint_geom_make(C.byref(g), 100, 154, 2) == 0
    assert (g.Hh, g.Wh) == (108, 164)           # roundup(100,8)+4, roundup(154,32)+4
    assert lib.nint_geom_make(C.byref(g), 0, 154, 2) == -1
    # packed weight image of the reference's first layer in bf16: (Cxp+Chp)*4*Ch16*k*k*2 bytes
    assert lib.nint_packed_weight_bytes(5, 64, 5, 1, 0) == (32 + 64) * 256 * 25 * 2
    # horizontal folding of thin first-layer inputs: pays when it lowers the x K-steps (reference layer 0: 25 -> 5)
    assert lib.nint_xfold_pays(5, 5, 1) == 1 and lib.nint_xfold_pays(5, 5, 0) == 1 and lib.nint_xfold_pays(4, 3, 0) == 1
    assert lib.nint_xfold_pays(62, 5, 1) == 0 and lib.nint_xfold_pays(5, 1, 1) == 0 and lib.nint_xfold_pays(64, 3, 1) == 0
    ly = _lib.NintLayer()
    ly.Cx, ly.Cxp, ly.Ch, ly.Ch16, ly.Chp, ly.k = 5, 32, 64, 64, 64, 5
    assert lib.nint_wgrad_workspace_bytes(C.byref(ly), 1, 256) > 0
    ly.k = 7
    assert lib.nint_wgrad_workspace_bytes(C.byref(ly), 1, 256) > 0    # 49 taps in two column groups
    ly.k = 9
    assert lib.nint_wgrad_workspace_bytes(C.byref(ly), 1, 256) == 0   # k=9 not instantiated


def test_product_refuses_cpu_tensors():
    import torch
    import nasa_niswan_amd as pkg
    net = pkg.ConvLSTM(4, [8], [3], 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 2, 4, 8, 8))
    with pytest.raises(AssertionError):
        pkg.ConvLSTM(4, [8, 8], [3], 1)          # model.py:237


def test_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """Error convention of the boundary: int return codes, never an abort.  Argument
    validation happens before any HIP call, so it can be exercised on a CPU-only machine."""
    from nasa_niswan_amd import _lib
    lib = _lib.load()
    g = _lib.NintGeom()
    assert lib.nint_geom_make(C.byref(g), 100, 154, 2) == 0
    ly = _lib.NintLayer()
    ly.Cx, ly.Cxp, ly.Ch, ly.Ch16, ly.Chp, ly.k = 5, 32, 64, 64, 64, 5
    E_ARG, E_SHAPE, E_ALIGN = -1, -2, -4
    assert lib.nint_cell_fwd(C.byref(ly), C.byref(g), 1, 8, None, None, None, None, None, None, None) == E_ARG
    assert lib.nint_cell_fwd(C.byref(ly), C.byref(g), 7, 8, 16, None, None, 16, 16, None, None) == E_ARG      # bad dtype
    assert lib.nint_cell_fwd(C.byref(ly), C.byref(g), 1, 8, 24, None, None, 16, 16, None, None) == E_ALIGN   # x slab not 16-B aligned
    ly.k = 4
    assert lib.nint_cell_fwd(C.byref(ly), C.byref(g), 1, 8, 16, None, None, 16, 16, None, None) == E_ARG      # even kernel (model.py:204)
    ly.k = 7
    assert lib.nint_cell_fwd(C.byref(ly), C.byref(g), 1, 8, 16, None, None, 16, 16, None, None) == E_ARG      # k/2 > physical halo
    ly.k = 5
    assert lib.nint_conv_dgrad(C.byref(ly), C.byref(g), 1, 8, None, None, None, None) == E_ARG
    assert lib.nint_conv_dgrad(C.byref(ly), C.byref(g), 1, 8, 16, None, None, None) == 0                     # nothing to produce: no-op
    assert lib.nint_cell_bwd_pointwise(C.byref(ly), C.byref(g), 1, 8, None, None, None, None, None, None, None, None) == E_ARG
    assert lib.nint_adam_flat(None, None, None, None, 10, 1e-3, 0.5, 0.999, 1e-8, 1, 1.0, None) == E_ARG
    assert lib.nint_adam_flat(16, 16, 16, 16, 10, 1e-3, 0.5, 0.999, 1e-8, 0, 1.0, None) == E_ARG             # step is 1-based
    assert lib.nint_loss_mse_l1_crop(16, 16, None, 16, None, 2, 1, 10, 10, 5, 5, 8, 8, None) == E_ARG         # crop outside the grid
    assert lib.nint_pack_btchw(16, 16, 2, 3, 5, 4, C.byref(g), 1, None) == E_ARG                              # Cp < C
    assert lib.nint_preproc_fuse_pad(None, None, 0, None, None, None, 1, 5, 5, 13, 13, 0, None) == E_ARG
    seq = _lib.NintSeq()
    assert lib.nint_seq_fwd(C.byref(seq), None) == E_ARG                                                      # L = 0
    assert lib.nint_seq_bwd(None, None) == E_ARG
    for code in (E_ARG, E_SHAPE, -3, E_ALIGN):
        assert lib.nint_error_string(code).decode().startswith("nint:")
    with pytest.raises(_lib.NintError, match="shape not supported"):
        _lib.check(E_SHAPE, "probe")


def test_modules_pickle_and_deepcopy_without_their_engines():
    import copy
    import pickle
    import torch
    import nasa_niswan_amd as pkg
    net = pkg.ConvLSTM(4, [8], [3], 1)
    net._engines["fake"] = object()
    twin = copy.deepcopy(net)
    assert twin._engines == {} and torch.equal(twin.conv.weight, net.conv.weight)
    again = pickle.loads(pickle.dumps(net))
    assert again._engines == {} and list(again.state_dict()) == list(net.state_dict())


def test_untrainable_kernel_sizes_are_known_before_any_backward():
    """The reference accepts any odd k (model.py:204); the weight-gradient kernel is instantiated for 1, 3, 5, 7.
    The engine learns that from the library at construction (pure host arithmetic, no GPU) and refuses a TRAINING
    workspace with a message that names the layer, instead of a generic shape error in the first backward()."""
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    assert SeqEngine.untrainable_layers([LayerCfg(5, 64, 5), LayerCfg(64, 32, 3), LayerCfg(32, 16, 1), LayerCfg(16, 8, 7)], "bf16") == []
    bad = SeqEngine.untrainable_layers([LayerCfg(5, 16, 3), LayerCfg(16, 8, 9)], "f32")
    assert len(bad) == 1 and "layer 1" in bad[0] and "k=9" in bad[0]


def test_library_reads_no_environment_and_owns_no_streams():
    """nint.h promises a stateless library: no hidden process-global switches.  The shared object must not import
    getenv, and must not create streams or events of its own (every launch goes to the caller's stream)."""
    from nasa_niswan_amd import _lib
    und = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    for sym in ("getenv", "secure_getenv", "hipStreamCreate", "hipStreamCreateWithFlags", "hipStreamCreateWithPriority",
                "hipEventCreate", "hipEventCreateWithFlags"):
        assert not re.search(rf"\b{sym}\b", und), f"libnint_hip.so imports {sym}"


def test_binding_picks_the_product_library_whatever_the_environment_says():
    """nothing in the environment can swap the library the package loads (round 2 had a NINT_LIB
    override for experiment builds; diagnostic tools now pass an explicit path to load())."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['NINT_LIB'] = '/nonexistent/libother.so';"
            "from nasa_niswan_amd import _lib; print(_lib.LIB_PATH); _lib.load()") % ROOT
    out = subprocess.check_output([sys.executable, "-c", code], text=True).strip()
    assert out.endswith(os.path.join("nasa-niswan_amd", "libnint_hip.so"))
    src = open(os.path.join(ROOT, "nasa-niswan_amd", "_lib.py")).read()
    assert "os.environ" not in src and "getenv" not in src


def test_graft_entry_build_passes():
    """The driver's build check: __graft_entry__.build() compiles (nothing to do when the library is current), imports the
    package and checks the library version against the binding's."""
    import __graft_entry__
    __graft_entry__.build()
