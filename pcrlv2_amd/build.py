"""Build libpcrl_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a pure C ABI.

    python -m pcrlv2_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libpcrl_hip.so")
OBJDIR = os.path.join(os.path.dirname(PKG), "build", "obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# Per-file code generation flags.  conv_brick16.hip: the plain wide-brick instantiations sit at the 256-register budget of two waves per SIMD; the greedy
# allocator's default assignment order spills 9-17 registers there (fragment addresses, reloaded behind s_waitcnt vmcnt(0) in the middle of a stage), the
# reverse order fits all of them (profiles/r05_b16_isa_mix.txt).
FILE_FLAGS = {"conv_brick16.hip": ["-mllvm", "-greedy-reverse-local-assignment"], "conv_brick16_bnr.hip": ["-mllvm", "-greedy-reverse-local-assignment"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "pcrl_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime()):
        return obj
    cmd = [hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[pcrlv2_amd.build] linked {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} objects")
    elif verbose:
        print(f"[pcrlv2_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
