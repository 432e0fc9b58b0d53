"""Build libudt_kernels.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU; the shared object lands next to this file so that it travels with
the repository snapshot to the GPU box.  Re-runs only what is out of date.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libudt_kernels.so")
SOURCES = ["api.hip", "gemm.hip", "lean.hip", "attention.hip", "tattn.hip", "norm.hip", "elementwise.hip", "pack.hip", "backward.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("UDT_EXTRA_FLAGS", "").split()


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm8.h"), os.path.join(CSRC, "conv3p.h"), os.path.join(CSRC, "lean.h"), os.path.join(CSRC, "wide.h"), os.path.join(CSRC, "rowres.h"), os.path.join(CSRC, "tile_common.h"), os.path.join(CSRC, "lean_params.h"), os.path.join(HERE, "..", "include", "udt_kernels.h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".o"))
    # an object is current only for the flags it was compiled with (a measurement build — UDT_EXTRA_FLAGS=-DUDT_MEASURE ... — must
    # never be linked into the product library because its objects happen to be newer than the sources)
    stamp, flags = o + ".flags", " ".join(FLAGS)
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags
    if not force and same_flags and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), _deps_mtime()):
        return o
    cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(flags)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return o


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[udifftext_amd.build] {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
