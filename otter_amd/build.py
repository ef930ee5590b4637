"""Build libotter_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m otter_amd.build [--force]

The library is a plain C-ABI shared object (include/otter_hip.h); it links no torch
symbols and is loaded with ctypes (otter_amd/_lib.py).
"""
import os
import re
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB = os.path.join(_PKG, "libotter_hip.so")
OBJDIR = os.path.join(_PKG, "csrc", "_obj")
SOURCES = ["common.hip", "index.hip", "encoder.hip", "gemm_bf16.hip"]
HEADERS = ["otter_common.h", "otter_device.h", "otter_gemm.h", os.path.join("..", "..", "include", "otter_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]
# Kernels that load into REGISTERS from inline asm and wait for them by hand (counted vmcnt): the register allocator
# does not know the destination is still in flight, so a spill there is not just slow -- it may reuse a register an
# asynchronous load is still going to write (seen once as a GPU memory fault).  The build fails if one needs scratch.
NO_SPILL = ("attn_full_kernel",)
# Hot kernels whose loads are LDS-DMA (no register destination): a spill is a performance bug, reported, not fatal.
WARN_SPILL = ("attn_pipe_kernel", "scan_kernel", "scan16_kernel", "stream_kernel", "stream16_kernel", "stream32_kernel", "gemm_pp_kernel",
              "merge_kernel")


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def check_no_spill(remarks, names=NO_SPILL):
    """Parse -Rpass-analysis=kernel-resource-usage remarks; return the kernels of `names` whose ScratchSize is not 0."""
    import re
    bad, name = [], None
    for ln in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
        if m and name is not None:
            if int(m.group(1)) > 0 and any(k in name for k in names):
                bad.append("%s (%s bytes/lane)" % (name, m.group(1)))
            name = None
    return bad


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
            res = open(obj + ".resources.txt", "w")
            procs.append((src, subprocess.Popen(cmd, stderr=res), res, obj))
    for src, p, res, obj in procs:
        rc = p.wait()
        res.close()
        text = open(obj + ".resources.txt").read()
        # warnings and errors only: drop the remarks and the source lines / carets quoted under them
        other, skip = [], False
        for ln in text.splitlines():
            if "remark:" in ln:
                skip = True
            elif re.search(r": (warning|error|note|fatal error):", ln):
                skip = False
            if not skip and ln.strip():
                other.append(ln)
        if other and verbose:
            print("\n".join(other), file=sys.stderr)
        if rc != 0:
            os.remove(obj) if os.path.exists(obj) else None
            raise RuntimeError("hipcc failed on %s" % src)
        bad = check_no_spill(text)
        if bad:
            os.remove(obj)
            raise RuntimeError("kernels with hand-ordered asynchronous register loads must not spill: %s" % ", ".join(bad))
        slow = check_no_spill(text, WARN_SPILL)
        if slow and verbose:
            print("[otter_amd.build] note: spills in hot kernels: %s" % ", ".join(slow), file=sys.stderr)
    if force or procs or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print("[otter_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
