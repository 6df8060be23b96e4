"""Builds fruits_amd/libfruits_hip.so (gfx950) with hipcc, in-tree.

``python -m fruits_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the built library travels with the tree.  The
walk kernel's template variants are spread over several objects that compile in
parallel.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libfruits_hip.so")
HEADERS = ["kernels.h", "plan.h", "jit.h", "walk.h", "walk_device.h", "walk_fused.h", "walk_types.h", "walk_scan.h", "coswiss.h", "walk_packed.h",
           "launch_cache.h", "static_programs.h",
           os.path.join("..", "..", "include", "fruits_hip.h")]


# headers that only some units include: {header: unit names}
LOCAL_HEADERS = {"pairwise.h": ("kernels_misc",)}


def units():
    """(object name, source, extra flags)"""
    out = [("kernels_misc", "kernels_misc.hip", []), ("plan", "plan.cpp", []),
           ("capi", "capi.cpp", []), ("jit", "jit.cpp", []),
           ("walk_static_reg", "walk_static_inst.hip", ["-DSTATIC_REGISTRY"])]
    # The walks: no a*b+c contraction - the reference rounds a letter's product before the
    # cumulative sum adds it (fruits/iss/semiring.py:143-149) - in EVERY unit, so that the record
    # interpreter (mode 0), the lean materialising walk (mode 2), the fused walk (mode 1), the
    # wave-per-series kernels and the static programs below agree bit for bit wherever they sum in
    # the same association (with contraction the interpreter fused the weight multiply of a
    # weighted plan's second scan with its first add: 1e-13 beside the others); and uniform
    # branches stay branches - structurised like divergent ones, every case of the fused walk's
    # level dispatch costs six scalar instructions and a speculative copy
    fused = ["-ffp-contract=off", "-mllvm", "-structurizecfg-skip-uniform-regions"]
    for mode in (0, 1, 2):
        for lv in (2, 4, 6, 8):
            out.append((f"walk_m{mode}_l{lv}", "walk_inst.hip",
                        [f"-DWALK_MODE={mode}", f"-DWALK_LV={lv}"] + fused))
    for lv in (2, 4, 6, 8):
        out.append((f"walk_m1ti_l{lv}", "walk_inst.hip",
                    ["-DWALK_MODE=1", f"-DWALK_LV={lv}", "-DWALK_TI"] + fused))
    for lv in (2, 4, 6, 8):
        out.append((f"walk_m1ho_l{lv}", "walk_inst.hip",
                    ["-DWALK_MODE=1", f"-DWALK_LV={lv}", "-DWALK_HO"] + fused))
    for mode in (0, 1):
        out.append((f"walk_packed_m{mode}", "walk_packed_inst.hip",
                    [f"-DWALK_MODE={mode}", "-ffp-contract=off"]))
    for s in (1, 2, 3, 4, 5, 6, 7, 8):
        out.append((f"coswiss_s{s}", "coswiss_inst.hip", [f"-DCOS_S={s}"]))
    for i in range(static_program_count()):
        # no a*b+c contraction: the interpreter cannot fuse a letter's product with the first
        # add of the scan (the product sits behind a branch on the letter's length) and the
        # reference rounds it too - the static kernels stay bit-identical to both
        out.append((f"walk_static_{i}", "walk_static_inst.hip",
                    [f"-DSTATIC_PROG={i}", "-ffp-contract=off"]))
    return out


def static_program_count() -> int:
    """Number of pre-generated static walk programs (csrc/static_programs.h, gen_static.py)."""
    with open(os.path.join(CSRC, "static_programs.h")) as f:
        for line in f:
            if line.startswith("// programs:"):
                return int(line.split(":")[1])
    return 0


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newest_header(unit: str = None) -> float:
    # (this file holds the units' compile flags: a change of it rebuilds like a header's)
    local = [h for h, us in LOCAL_HEADERS.items() if unit is None or unit in us]
    return max([os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + local]
               + [os.path.getmtime(__file__)])


STAMP = LIB + ".flags"


def build_tag() -> str:
    """What distinguishes two builds of the same sources: the compile flags that
    come from the environment (the diagnostic timing build carries stamps and
    early returns and must never be mistaken for the product library)."""
    return "timing" if os.environ.get("FRUITS_HIP_TIMING_BUILD") else "product"


def _stamped_tag() -> str:
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return ""


def needs_build() -> bool:
    if not os.path.exists(LIB) or _stamped_tag() != build_tag():
        return True
    t = os.path.getmtime(LIB)
    srcs = {os.path.join(CSRC, u[1]) for u in units()}
    return any(os.path.getmtime(s) > t for s in srcs) or _newest_header() > t


def build_native(force: bool = False, verbose: bool = False, jobs: int = 0,
                 variant: str = "", defines=(), only=()) -> str:
    """``variant`` / ``defines``: an experiment library next to the product one
    (``libfruits_hip.<variant>.so``, selected at run time with FRUITS_HIP_LIB), built
    with extra ``-D`` flags; its objects carry the variant in their names.  ``only``:
    unit names rebuilt with the flags - the rest are the product's objects."""
    if variant:
        return _build(LIB[:-3] + f".{variant}.so", f"v{variant}", list(defines), True, verbose, jobs,
                      only=tuple(only))
    if not force and not needs_build():
        return LIB
    tag = "t" if os.environ.get("FRUITS_HIP_TIMING_BUILD") else "p"
    lib = _build(LIB, tag, [], force, verbose, jobs)
    with open(STAMP, "w") as f:
        f.write(build_tag() + "\n")
    return lib


JIT_HEADERS = ["walk_types.h", "walk_scan.h", "walk_device.h", "walk_fused.h"]


def write_jit_sources() -> str:
    """csrc/jit_sources.inc: the device headers as ONE string literal (local includes and
    include guards dropped) - the text jit.cpp hands to hipRTC in front of a schedule."""
    parts = []
    for h in JIT_HEADERS:
        with open(os.path.join(CSRC, h)) as f:
            for line in f:
                if line.startswith('#include "') or line.startswith("#pragma once"):
                    continue
                parts.append(line)
    text = "".join(parts)
    assert ')FRJIT"' not in text
    out = os.path.join(CSRC, "jit_sources.inc")
    body = "static const char *kJitDeviceSource = R\"FRJIT(\n" + text + ")FRJIT\";\n"
    try:
        with open(out) as f:
            if f.read() == body:
                return out
    except OSError:
        pass
    with open(out, "w") as f:
        f.write(body)
    return out


def _build(lib_path: str, tag: str, defines, force: bool, verbose: bool, jobs: int,
           only=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    write_jit_sources()
    cc = hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
             "-Wno-unused-function"] + [f"-D{d}" for d in defines]
    if os.environ.get("FRUITS_HIP_TIMING_BUILD"):
        flags.append("-DFRUITS_HIP_TIMING_BUILD")
    def compile_one(unit):
        name, src, extra = unit
        hdr_t = _newest_header(name)
        srcp = os.path.join(CSRC, src)
        if only and name not in only:
            return os.path.join(OBJ, f"{name}.p.o")      # the product's object
        obj = os.path.join(OBJ, f"{name}.{tag}.o")
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_t)):
            return obj
        cmd = [cc] + flags + extra + ["-x", "hip", "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    jobs = jobs or min(8, os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, units()))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path + ".tmp"] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    variant, defines, only = "", [], []
    for arg in sys.argv[1:]:
        if arg.startswith("--variant="):
            variant = arg.split("=", 1)[1]
        elif arg.startswith("--only="):
            only = arg.split("=", 1)[1].split(",")
        elif arg.startswith("-D"):
            defines.append(arg[2:])
    print(build_native(force="--force" in sys.argv, verbose="-v" in sys.argv, variant=variant,
                       defines=defines, only=only))
    if not variant and "--no-bundle" not in sys.argv:
        from fruits_amd.gen_bundle import build_bundle
        print(build_bundle(force="--force" in sys.argv))
