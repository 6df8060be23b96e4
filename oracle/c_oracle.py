"""ctypes wrapper of oracle/iss_oracle.c (TEST / BASELINE INFRASTRUCTURE ONLY).

Never imported from ``fruits_amd``.  ``build()`` compiles the library with gcc;
the tests, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` are the only users.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import ref_numpy as _np_orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "iss_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_iterated_sum_fast.restype = C.c_int
        _lib.orc_iss_batch.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def iterated_sum_fast(Z, word, alpha, lookup, extended, total, nthreads=0, semiring="Reals"):
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    word = np.ascontiguousarray(word, dtype=np.int32)
    alpha = np.ascontiguousarray(alpha, dtype=np.float32)
    N, D, T = Z.shape
    L, Dw = word.shape
    out = np.zeros((N, extended, T))
    lk = None
    if lookup is not None:
        lk = np.ascontiguousarray(lookup, dtype=np.float64)
    rc = lib().orc_iterated_sum_fast(
        _p(Z, _dp), C.c_int64(N), C.c_int64(D), C.c_int64(T), _p(word, _ip),
        C.c_int(L), C.c_int(Dw), _p(alpha, _fp),
        _p(lk, _dp) if lk is not None else None, C.c_int64(extended),
        C.c_int((1 if total else 0) | (2 if semiring == "Arctic" else 0) | (4 if semiring == "Bayesian" else 0)), _p(out, _dp),
        C.c_int(nthreads))
    if rc != 0:
        raise ValueError("orc_iterated_sum_fast: bad arguments")
    return out


def iss_transform(X, word_strings, mode="SINGLE", alphas=None, lookup=None,
                  total=False, nthreads=0, out=None, semiring="Reals"):
    """Same contract as ref_numpy.iss_transform -> (K, N, T)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, D, T = X.shape
    plan = (_np_orc.cache_plan(word_strings) if mode == "EXTENDED"
            else [1] * len(word_strings))
    rows = [np.array(_np_orc.parse_word(s), dtype=np.int32) for s in word_strings]
    W = len(rows)
    exps = np.concatenate([r.ravel() for r in rows]).astype(np.int32)
    word_off = np.zeros(W, dtype=np.int64)
    alpha_off = np.zeros(W, dtype=np.int64)
    Ls = np.array([r.shape[0] for r in rows], dtype=np.int32)
    Dws = np.array([r.shape[1] for r in rows], dtype=np.int32)
    o = a = 0
    for i, r in enumerate(rows):
        word_off[i], alpha_off[i] = o, a
        o += r.size
        a += r.shape[0]
    if lookup is None:
        alpha = None
        lk = None
        total = True
    else:
        alpha = np.concatenate([
            np.ones(r.shape[0], dtype=np.float32) if (alphas is None or alphas[i] is None)
            else np.asarray(alphas[i], dtype=np.float32)
            for i, r in enumerate(rows)]).astype(np.float32)
        lk = np.ascontiguousarray(lookup, dtype=np.float64)
        assert lk.shape[0] >= N and lk.shape[1] == T
    K = int(sum(plan))
    if out is None:
        out = np.zeros((K, N, T))
    depth = np.array(plan, dtype=np.int32)
    rc = lib().orc_iss_batch(
        _p(X, _dp), C.c_int64(N), C.c_int64(D), C.c_int64(T), C.c_int(W),
        _p(exps, _ip), _p(word_off, _lp), _p(Ls, _ip), _p(Dws, _ip),
        _p(alpha, _fp) if alpha is not None else None, _p(alpha_off, _lp),
        _p(depth, _ip), _p(lk, _dp) if lk is not None else None,
        C.c_int((1 if total else 0) | (2 if semiring == "Arctic" else 0) | (4 if semiring == "Bayesian" else 0)), _p(out, _dp),
        C.c_int(nthreads))
    if rc != 0:
        raise ValueError("orc_iss_batch: word dimension exceeds input dimension")
    return out


def increments(X, k=1):
    X = np.ascontiguousarray(X, dtype=np.float64)
    out = np.empty_like(X)
    T = X.shape[-1]
    lib().orc_increments(_p(X, _dp), C.c_int64(X.size // T), C.c_int64(T),
                         C.c_int64(k), _p(out, _dp))
    return out


def l1_sum(X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, D, T = X.shape
    out = np.empty((N, T))
    lib().orc_l1_sum(_p(X, _dp), C.c_int64(N), C.c_int64(D), C.c_int64(T), _p(out, _dp))
    return out


def _sieve(fn, A, cuts, q):
    A = np.ascontiguousarray(A, dtype=np.float64)
    cuts = np.ascontiguousarray(cuts, dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    N, T = A.shape
    out = np.zeros((N, (cuts.shape[1] - 1) * (len(q) - 1)))
    fn(_p(A, _dp), C.c_int64(N), C.c_int64(T), _p(cuts, _lp), C.c_int(cuts.shape[1]),
       _p(q, _dp), C.c_int(len(q)), _p(out, _dp))
    return out


def npi_backend(A, cuts, q):
    return _sieve(lib().orc_npi_backend, A, cuts, q)


def mpi_backend(A, cuts, q):
    return _sieve(lib().orc_mpi_backend, A, cuts, q)


def end_transform(A, cuts):
    A = np.ascontiguousarray(A, dtype=np.float64)
    cuts = np.ascontiguousarray(cuts, dtype=np.int64)
    N, T = A.shape
    out = np.zeros((N, cuts.shape[1] - 1))
    lib().orc_end(_p(A, _dp), C.c_int64(N), C.c_int64(T), _p(cuts, _lp),
                  C.c_int(cuts.shape[1]), _p(out, _dp))
    return out
