"""Kernels shipped with the build: ``python -m fruits_amd.gen_bundle`` (run by
``fruits_amd.build`` / ``__graft_entry__.build()``) compiles - hipRTC, no GPU needed - the fused
kernels that the pipelines of the reference's experiment fruits and of the BASELINE configs would
otherwise compile on their first run on a machine (csrc/jit.cpp: the sieves as immediates, a small
plan as straight-line code, a large one in pieces) into ``fruits_amd/jit_bundle`` in the run-time
cache's own file format.  The library looks there behind the user's cache and in front of the
compiler (``fr_pipeline_prepare_cached``: milliseconds), so the FIRST ``Fruit.transform`` of such
a fruit on a fresh machine already runs its own kernels.

What a kernel depends on is the slice's word set / semiring / weighting mode (the plan), the
sieves (kind, differencing order, cuts - those depend on the series length T) and the kind of the
thresholds (which are infinite); the thresholds themselves, the batch size, the input's
dimensions and the preparateurs are run-time arguments.

The manifest: the first slices of experiments/fruit_reduced.py, fruit_general.py, fruit_twi.py
(the Reals slices - where a fruit's transform spends its time; the CosWISS kernels are compiled
ahead of time anyway, the Arctic chains compile in seconds at run time) at the series lengths of
the BASELINE configs, and the bench pipelines of configs 3 / 4 / 5.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(HERE, "jit_bundle")
STAMP = os.path.join(BUNDLE, "MANIFEST")


def _experiment_sieves(fr):
    S = fr.sieving
    return [S.NPI(q=(0.5, 1.0), inc=0), S.NPI(q=(0.5, 1.0), inc=1), S.NPI(q=(0.5, 1.0), inc=2),
            S.MPI(q=(0.5, 1.0), inc=0), S.MPI(q=(0.5, 1.0), inc=1), S.MPI(q=(0.5, 1.0), inc=2), S.END()]


def manifest(fr):
    """(name, T, ISS, sieves, groups per series of a straight-line plan)."""
    W, E = fr.words, fr.ISSMode.EXTENDED
    Idx, L1 = fr.iss.weighting.Indices, fr.iss.weighting.L1
    Arctic = fr.iss.semiring.Arctic
    S = fr.sieving

    def alt(*strings):
        return W.alternate_sign([W.SimpleWord(s) for s in strings])
    return [
        # the bench pipelines (bench.py: INC -> ISS -> NPI(q=(0.5, 1)), END)
        ("config 3", 1024, fr.ISS(W.of_weight(4, 2), mode=E, weighting=Idx()),
         [S.NPI(q=(0.5, 1.0)), S.END()], (1,)),
        ("config 4", 1024, fr.ISS(W.of_weight(6, 2), mode=E, weighting=Idx()),
         [S.NPI(q=(0.5, 1.0)), S.END()], (1,)),
        ("config 5", 4096, fr.ISS(W.of_weight(9, 1), mode=E, weighting=L1()), [S.NPI(), S.END()], (1,)),
        # experiments/fruit_reduced.py:27-39 (slice 1)
        ("fruit_reduced 1", 1024, fr.ISS(W.of_weight(4, 2), mode=E, weighting=Idx()),
         _experiment_sieves(fr), (1,)),
        # experiments/fruit_general.py (slice 1)
        ("fruit_general 1", 1024, fr.ISS(W.of_weight(6, 2), mode=E, weighting=Idx()),
         _experiment_sieves(fr), (1,)),
        # experiments/fruit_twi.py:5-17 (slice 1)
        ("fruit_twi 1", 4096, fr.ISS(W.of_weight(9, 1), mode=E, weighting=L1()),
         [S.NPI(), S.MPI(), S.END()], (1,)),
    ]


def placeholder_quantiles(sieves, K: int, q_stride: int) -> np.ndarray:
    """A (K, q_stride) threshold table with the infinities where fitted thresholds have them
    (SegmentSieve._fit: q = 1 -> +inf, -1 -> -inf, 0 -> 0) and the quantile LEVEL where a fitted
    value goes: every finite value stands for any other in a kernel's source."""
    from .sieving.segment import END
    row, off = np.zeros(q_stride), 0
    for sv in sieves:
        if type(sv) is END:
            continue
        qs = np.sort([np.inf if q == 1.0 else (-np.inf if q == -1.0 else float(q)) for q in sv._q])
        row[off:off + len(qs)] = qs
        off += len(qs)
    return np.tile(row, (K, 1))


def bundle_slice(entry) -> tuple:
    """Compiles the kernels of one manifest entry into the bundle; (name, code objects, seconds)."""
    import fruits_amd as fr
    from fruits_amd import _native as nat
    from fruits_amd.fruit import FruitSlice
    name, T, iss, sieves, groups = entry
    t0 = time.time()
    plan = iss._plan(0, len(iss.words))
    specs, _, _ = FruitSlice._pipeline_specs(sieves, T)
    pipe = nat.Pipeline(plan, specs, T)
    quant = placeholder_quantiles(sieves, plan.rows, pipe.q_stride)
    n = 0
    for g in groups:
        n = max(n, pipe.bundle(quant, BUNDLE, groups=g))
    return name, n, time.time() - t0


def manifest_key() -> str:
    """What the bundle depends on: the device sources the kernels are generated from, this file,
    the plan compiler and the JIT's source generator."""
    h = hashlib.sha256()
    for f in ("csrc/walk_types.h", "csrc/walk_scan.h", "csrc/walk_device.h", "csrc/walk_fused.h",
              "csrc/jit.cpp", "csrc/plan.cpp", "csrc/capi.cpp", "gen_bundle.py"):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def up_to_date() -> bool:
    try:
        with open(STAMP) as f:
            return f.read().split()[0] == manifest_key()
    except (OSError, IndexError):
        return False


def build_bundle(force: bool = False, jobs: int = 0, verbose: bool = True) -> str:
    """Builds fruits_amd/jit_bundle (skipped when it is current).  One process per manifest entry
    (a large plan compiles its piece types on threads of its own)."""
    if not force and up_to_date():
        return BUNDLE
    if force and os.path.isdir(BUNDLE):
        for f in os.listdir(BUNDLE):
            os.remove(os.path.join(BUNDLE, f))
    os.makedirs(BUNDLE, mode=0o700, exist_ok=True)
    # (incremental: a code object's file name hashes its source, so what an earlier build left
    # and is still wanted is found - and touched - instead of compiled again; the rest is swept)
    started = time.time() - 2.0
    import fruits_amd as fr
    entries = manifest(fr)
    # the entries with the most to compile first (nodes x sieves); every entry compiles its
    # kernels on threads of its own (csrc/capi.cpp, fr_pipeline_bundle) - together a little more
    # than the cores, the compiler's longest single job (~2 min here) is what bounds the build
    entries.sort(key=lambda e: -len(e[2].words) * len(e[3]))
    cores = os.cpu_count() or 2
    jobs = jobs or min(len(entries), max(1, cores // 2))
    os.environ.setdefault("FRUITS_BUNDLE_THREADS", str(max(2, cores // 2)))
    os.environ["FRUITS_HIP_JIT_CACHE"] = ""      # (nothing of this goes through the user's cache)
    t0 = time.time()
    lines = []
    with cf.ProcessPoolExecutor(max_workers=jobs) as ex:
        for name, n, dt in ex.map(bundle_slice, entries):
            lines.append(f"{name}: {n} code objects, {dt:.0f} s")
            if verbose:
                print(f"[bundle] {lines[-1]}", flush=True)
    for f in os.listdir(BUNDLE):
        if f.endswith(".co") and os.path.getmtime(os.path.join(BUNDLE, f)) < started:
            os.remove(os.path.join(BUNDLE, f))
    files = sorted(f for f in os.listdir(BUNDLE) if f.endswith(".co"))
    with open(STAMP, "w") as f:
        f.write(manifest_key() + "\n" + "\n".join(lines) + "\n" + "\n".join(files) + "\n")
    if verbose:
        size = sum(os.path.getsize(os.path.join(BUNDLE, x)) for x in files)
        print(f"[bundle] {len(files)} code objects, {size / 1e6:.1f} MB, {time.time() - t0:.0f} s", flush=True)
    return BUNDLE


if __name__ == "__main__":
    build_bundle(force="--force" in sys.argv)
