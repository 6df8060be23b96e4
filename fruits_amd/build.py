"""Builds fruits_amd/libfruits_hip.so (gfx950) with hipcc, in-tree.

``python -m fruits_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the built library travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfruits_hip.so")
SOURCES = ["iss_kernels.hip", "plan.cpp", "capi.cpp"]
HEADERS = ["kernels.h", "plan.h", os.path.join("..", "..", "include", "fruits_hip.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + (["-DFRUITS_HIP_TIMING_BUILD"] if os.environ.get("FRUITS_HIP_TIMING_BUILD") else []) + [
           "-Wall", "-Wno-unused-function", "-o", LIB + ".tmp"]
    cmd += ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
