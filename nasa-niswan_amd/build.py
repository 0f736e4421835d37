"""Build recipe for the C-ABI HIP library (gfx950 only).

    python nasa-niswan_amd/build.py            # -> nasa-niswan_amd/libnint_hip.so

hipcc cross-compiles for gfx950 without a GPU; the .so is built in-tree so that it travels
to the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libnint_hip.so")
SOURCES = ["conv_igemm.hip", "wgrad.hip", "pointwise.hip", "seq.hip"]
EXP_SOURCES = ["conv_ws.hip"]      # experiment build only (tools/kbench.py --exp --dbg 0x1000): measured, not shipped
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-variable", "-Wno-unused-local-typedef"]


def _newer(a, bs):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(b) for b in bs)


def build(force: bool = False, verbose: bool = False, exp: bool = False) -> str:
    """exp=True: the EXPERIMENT build (-DNINT_EXPERIMENT: extra tile configurations selectable through
    the upper bits of nint_layer.tile_rows for tools/kbench.py) into libnint_hip_exp.so; never loaded by the package itself."""
    global OBJ, LIB, FLAGS
    if exp:
        OBJ, LIB = os.path.join(HERE, "build_exp"), os.path.join(HERE, "libnint_hip_exp.so")
        FLAGS = FLAGS + ["-DNINT_EXPERIMENT"]
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "nint_common.h"), os.path.join(ROOT, "include", "nint.h")]
    objs, jobs = [], []
    for src in SOURCES + (EXP_SOURCES if exp else []):
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + headers):
            jobs.append([HIPCC, *FLAGS, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, exp="--exp" in sys.argv))
