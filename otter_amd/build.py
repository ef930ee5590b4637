"""Build libotter_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m otter_amd.build [--force]

The library is a plain C-ABI shared object (include/otter_hip.h); it links no torch
symbols and is loaded with ctypes (otter_amd/_lib.py).
"""
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB = os.path.join(_PKG, "libotter_hip.so")
OBJDIR = os.path.join(_PKG, "csrc", "_obj")
SOURCES = ["common.hip", "index.hip", "encoder.hip", "gemm_bf16.hip"]
HEADERS = ["otter_common.h", "otter_device.h", "otter_gemm.h", os.path.join("..", "..", "include", "otter_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build_lib(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_time = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(sp), hdr_time):
            cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print("[otter_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[otter_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
