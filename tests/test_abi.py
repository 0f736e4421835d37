"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/nint.h declares, and the ctypes structures match the header's layout.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

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
    assert lib.nint_version() == 101
    assert lib.nint_kc(0) == 16 and lib.nint_kc(1) == 32
    assert lib.nint_error_string(-2).decode().startswith("nint:")


def test_struct_layout_matches_header(tmp_path):
    """Compile a tiny C program against nint.h and compare sizeof/offsetof with ctypes."""
    from nasa_niswan_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nint.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(nint_geom),sizeof(nint_layer),sizeof(nint_seq),offsetof(nint_layer,Wf),offsetof(nint_seq,xs),'
                    'offsetof(nint_seq,dW),offsetof(nint_seq,wg_partial_bytes));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.NintGeom), C.sizeof(_lib.NintLayer), C.sizeof(_lib.NintSeq), _lib.NintLayer.Wf.offset,
            _lib.NintSeq.xs.offset, _lib.NintSeq.dW.offset, _lib.NintSeq.wg_partial_bytes.offset]
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
    ly = _lib.NintLayer()
    ly.Cx, ly.Cxp, ly.Ch, ly.Ch16, ly.Chp, ly.k = 5, 32, 64, 64, 64, 5
    assert lib.nint_wgrad_workspace_bytes(C.byref(ly), 1, 256) > 0
    ly.k = 7
    assert lib.nint_wgrad_workspace_bytes(C.byref(ly), 1, 256) == 0   # k=7 not instantiated


def test_product_refuses_cpu_tensors():
    import torch
    import nasa_niswan_amd as pkg
    net = pkg.ConvLSTM(4, [8], [3], 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 2, 4, 8, 8))
    with pytest.raises(AssertionError):
        pkg.ConvLSTM(4, [8, 8], [3], 1)          # model.py:237
