"""ctypes binding of the C ABI declared in include/nint.h.

The HIP library is the product: if ``libnint_hip.so`` is missing or cannot be loaded this
module raises -- there is no CPU or eager-PyTorch fallback anywhere in the package."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnint_hip.so")     # the one product library; nothing in the environment can swap it

NINT_F32, NINT_BF16 = 0, 1
NINT_OK, NINT_E_ARG, NINT_E_SHAPE, NINT_E_LDS, NINT_E_ALIGN = 0, -1, -2, -3, -4
NINT_MAX_LAYERS = 8
NINT_VERSION = 111     # include/nint.h NINT_VERSION: the library this binding was written against
NINT_LOSS_SCRATCH_FLOATS = 8194
NINT_LOSS_STATS = 8

vp = C.c_void_p


class NintGeom(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("P", C.c_int32), ("Hh", C.c_int32), ("Wh", C.c_int32)]


class NintLayer(C.Structure):
    _fields_ = [("Cx", C.c_int32), ("Cxp", C.c_int32), ("Ch", C.c_int32), ("Ch16", C.c_int32), ("Chp", C.c_int32),
                ("k", C.c_int32), ("tile_rows", C.c_int32), ("xfold", C.c_int32), ("wide", C.c_int32),
                ("Wf", vp), ("Wd", vp), ("bias_p", vp)]


class NintSeq(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("T", C.c_int32), ("L", C.c_int32),
                ("need_dx", C.c_int32), ("has_init_state", C.c_int32), ("n_cu", C.c_int32), ("zero_dstate", C.c_int32),
                ("g", NintGeom), ("layer", NintLayer * NINT_MAX_LAYERS),
                ("xs", vp), ("h", vp * NINT_MAX_LAYERS), ("c", vp * NINT_MAX_LAYERS),
                ("gates", vp * NINT_MAX_LAYERS), ("dG", vp * NINT_MAX_LAYERS), ("dh", vp * NINT_MAX_LAYERS),
                ("dc", vp * NINT_MAX_LAYERS), ("dx", vp), ("dW", vp * NINT_MAX_LAYERS), ("db", vp * NINT_MAX_LAYERS),
                ("wg_partial", vp), ("wg_partial_bytes", C.c_size_t), ("fuse_bwd", C.c_int32),
                ("probe_mask", C.c_int32), ("probe", vp), ("probe_slots", C.c_int32),
                ("wave", C.c_int32), ("bwd_parts", C.c_int32)]


# every symbol include/nint.h declares: name -> (restype, argtypes)
_I, _SZ, _F = C.c_int, C.c_size_t, C.c_float
_PG, _PL, _PS = C.POINTER(NintGeom), C.POINTER(NintLayer), C.POINTER(NintSeq)
SIGNATURES = {
    "nint_version": (_I, []),
    "nint_error_string": (C.c_char_p, [_I]),
    "nint_device_info": (_I, [C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    "nint_selftest": (_I, [vp, vp]),
    "nint_kc": (_I, [_I]),
    "nint_geom_make": (_I, [_PG, _I, _I, _I]),
    "nint_pack_btchw": (_I, [vp, vp, _I, _I, _I, _I, _PG, _I, vp]),
    "nint_unpack_halo": (_I, [vp, vp, _I, _I, _I, _I, _PG, _I, vp]),
    "nint_pack_compact": (_I, [vp, vp, _I, _I, _I, _I, _I, _I, vp]),
    "nint_unpack_compact": (_I, [vp, vp, _I, _I, _I, _I, _I, _I, vp]),
    "nint_packed_weight_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "nint_xfold_pays": (_I, [_I, _I, _I]),
    "nint_pack_weights": (_I, [vp, vp, vp, vp, vp, _I, _I, _I, _I, _I, vp]),
    "nint_pack_weights_layers": (_I, [C.POINTER(vp), C.POINTER(vp), C.POINTER(NintLayer), _I, _I, vp]),
    "nint_pack_btchw_xfold": (_I, [vp, vp, _I, _I, _I, _I, _I, _PG, _I, vp]),
    "nint_unfold_dx": (_I, [vp, vp, _I, _I, _I, _I, _I, _I, _I, vp]),
    "nint_stencil_holds": (_I, [_PL]),
    "nint_cell_fwd": (_I, [_PL, _PG, _I, _I, vp, vp, vp, vp, vp, vp, vp]),
    "nint_cell_bwd_pointwise": (_I, [_PL, _PG, _I, _I, vp, vp, vp, vp, vp, vp, vp]),
    "nint_conv_dgrad": (_I, [_PL, _PG, _I, _I, vp, vp, vp, vp]),
    "nint_cell_bwd_fused": (_I, [_PL, _PG, _I, _I, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nint_wgrad_workspace_bytes": (_SZ, [_PL, _I, _I]),
    "nint_conv_wgrad": (_I, [_PL, _PG, _I, _I, vp, vp, vp, vp, vp, vp, _SZ, _I, vp]),
    "nint_seq_fwd": (_I, [_PS, vp]),
    "nint_seq_bwd": (_I, [_PS, vp]),
    "nint_head_fwd": (_I, [vp, _I, _I, _I, _I, _I, vp, vp, vp, _PG, _I, vp]),
    "nint_head_bwd": (_I, [vp, _I, _I, _I, _I, _I, vp, vp, vp, vp, vp, _PG, _I, vp, _SZ, vp]),
    "nint_loss_mse_l1_crop": (_I, [vp, vp, vp, vp, vp, _I, _I, _I, _I, _I, _I, _I, _I, vp]),
    "nint_head_loss_fused": (_I, [vp, _I, _I, _I, _I, _I, vp, vp, vp, vp, vp, vp, vp, _PG, _I, _I, _I, _I, _I, vp]),
    "nint_adam_flat": (_I, [vp, vp, vp, vp, _SZ, C.c_double, C.c_double, C.c_double, C.c_double, _I, _F, vp]),
    "nint_preproc_fuse_pad": (_I, [C.POINTER(vp), C.POINTER(_I), _I, vp, vp, vp, _I, _I, _I, _I, _I, _I, vp]),
    "nint_preproc_fuse_pad_batch": (_I, [C.POINTER(vp), C.POINTER(_I), _I, vp, vp, C.POINTER(_I), _I, vp, _I, _I, _I, _I, _I, _I, vp]),
    "nint_preproc_fuse_pad_slab": (_I, [C.POINTER(vp), C.POINTER(_I), _I, vp, vp, C.POINTER(_I), _I, vp, _I, _I, _I, _I, _I, _PG, _I, _I, vp]),
}

_lib = None


class NintError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """Load the HIP library (after torch, so that both share one HIP runtime) and bind signatures.
    ``path`` is the only way to pick another build (diagnostic tools load a -DNINT_STAMP build this way, before
    anything else touches the package); the first successful call wins for the life of the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise NintError(f"{path} not found: build it with `python nasa-niswan_amd/build.py` "
                        "(the HIP extension is mandatory; there is no fallback path)")
    import torch  # noqa: F401  -- loads torch's bundled libamdhip64.so.7 first; ours binds to the same SONAME
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.nint_version() != NINT_VERSION:
        raise NintError("libnint_hip.so version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().nint_error_string(rc).decode()
        raise NintError(f"{what or 'nint call'} failed: {msg} (code {rc})")


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
