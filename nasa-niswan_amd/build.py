"""Build recipe for the C-ABI HIP library (gfx950 only).

    python nasa-niswan_amd/build.py            # -> nasa-niswan_amd/libnint_hip.so

hipcc cross-compiles for gfx950 without a GPU; the .so is built in-tree so that it travels
to the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).

Diagnostic variants (phase stamps for tools/clockprobe.py, A/B copies for tools/ab.py) are built by
``build(out=..., defines=[...])`` into a path of the caller's choice; the package itself only ever loads
libnint_hip.so."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libnint_hip.so")
SOURCES = ["conv_igemm.hip", "stencil.hip", "tiny_gemm.hip", "wgrad.hip", "pointwise.hip", "seq.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-variable", "-Wno-unused-local-typedef"]


def _newer(a, bs):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(b) for b in bs)


def build(force: bool = False, verbose: bool = False, out: Optional[str] = None, defines: Sequence[str] = (),
          csrc: Optional[str] = None) -> str:
    """Compile SOURCES and link them into `out` (default: the product library).  `defines` (e.g. ["NINT_STAMP"]) and
    `csrc` (another source directory with the same file names: an A/B copy) select a diagnostic variant, whose objects go
    to a directory of their own next to `out`."""
    lib_path = out or LIB
    src_dir = csrc or CSRC
    variant = bool(out or defines or csrc)
    if variant and not out:
        raise ValueError("a diagnostic variant needs its own output path")
    obj_dir = (os.path.splitext(lib_path)[0] + "_obj") if variant else OBJ
    flags = FLAGS + ["-D" + d for d in defines]
    if csrc:
        flags = [f for f in flags if f != "-I" + CSRC] + ["-I" + src_dir]
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(src_dir, "nint_common.h"), os.path.join(ROOT, "include", "nint.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(src_dir, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + headers):
            jobs.append([HIPCC, *flags, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=5) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not _newer(lib_path, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
    return lib_path


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    srcs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--csrc=")]
    print(build(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None, defines=defs,
                csrc=srcs[0] if srcs else None))
