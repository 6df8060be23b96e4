"""ctypes binding of libfruits_hip.so (the C ABI of include/fruits_hip.h).

This is the only door between the Python host code and the HIP kernels.  There
is NO CPU fallback: if the library is missing or no HIP device is present every
compute entry raises.  PyTorch is used for plumbing only - device memory
(``torch.empty(..., device="cuda")``), the current HIP stream and, in
``fruits_amd.parallel``, ``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRUITS_HIP_LIB: an experiment build of the same sources (fruits_amd.build --variant=...)
LIB_PATH = os.path.join(_HERE, os.environ.get("FRUITS_HIP_LIB", "libfruits_hip.so"))

FR_W_NONE, FR_W_NONTOTAL, FR_W_TOTAL = 0, 1, 2
FR_SIEVE_NPI, FR_SIEVE_MPI, FR_SIEVE_END = 0, 1, 2
FR_SIEVE_SERIES_CUTS = 0x100   # OR-ed into a kind: the sieve's cuts are slots of a per-series table
(FR_INFO_ROWS, FR_INFO_NODES, FR_INFO_LEVELS, FR_INFO_DIMS_USED, FR_INFO_MAX_DIM,
 FR_INFO_ALPHAS, FR_INFO_GROUPS, FR_INFO_SHARED, FR_INFO_STAGED_ROWS, FR_INFO_JIT_PROGRAMS,
 FR_INFO_AOT_PROGRAM) = range(11)
FR_E_ARG, FR_E_DIM, FR_E_HIP, FR_E_NOMEM, FR_E_LIMIT, FR_E_INDEX = -1, -2, -3, -4, -5, -6

EXPORTS = [
    "fr_last_error", "fr_version", "fr_device_count", "fr_malloc", "fr_free",
    "fr_memcpy_h2d", "fr_memcpy_d2h", "fr_stream_sync", "fr_plan_create",
    "fr_plan_destroy", "fr_plan_info", "fr_plan_dump", "fr_plan_records", "fr_plan_static_schedule", "fr_plan_pieces", "fr_plan_jit", "fr_plan_workspace_bytes",
    "fr_iss_run", "fr_iterated_sum_fast_host", "fr_increments",
    "fr_pathlen_lookup", "fr_sieve", "fr_pre_transform", "fr_standardize",
    "fr_pipeline_create", "fr_pipeline_destroy", "fr_pipeline_info",
    "fr_pipeline_workspace_bytes", "fr_pipeline_run", "fr_pipeline_set_quantiles",
    "fr_select_ranks", "fr_select_ranks_begin", "fr_select_ranks_end", "fr_coswiss_combine", "fr_plan_create_coswiss", "fr_nan_to_num",
    "fr_plan_prepare", "fr_pipeline_prepare", "fr_pipeline_compile_plan", "fr_pipeline_prepare_cached", "fr_pipeline_bundle", "fr_plan_fits", "fr_release_scratch",
    "fr_pipeline_set_preparation", "fr_pipeline_set_series_cuts", "fr_pipeline_set_argmax", "fr_arctic_argmax", "fr_coswiss_set_dropout",
    "fr_coswiss_set_input_stride", "fr_coswiss_ffn",
]

_lib = None
_torch = None


class NativeError(RuntimeError):
    pass


def torch():
    """torch is imported before the library so both share ONE HIP runtime
    (torch bundles libamdhip64.so.7; the loader resolves our DT_NEEDED entry to
    the copy that is already mapped)."""
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "fruits_amd/libfruits_hip.so is missing - build it with "
            "`python -m fruits_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    torch()
    L = C.CDLL(LIB_PATH)
    L.fr_last_error.restype = C.c_char_p
    L.fr_plan_create.restype = C.c_void_p
    L.fr_plan_create_coswiss.restype = C.c_void_p
    L.fr_plan_destroy.restype = None
    L.fr_plan_destroy.argtypes = [C.c_void_p]
    L.fr_plan_info.restype = C.c_int64
    L.fr_plan_info.argtypes = [C.c_void_p, C.c_int32]
    L.fr_plan_dump.restype = C.c_int32
    L.fr_plan_records.restype = C.c_int32
    L.fr_plan_jit.restype = C.c_int32
    L.fr_plan_jit.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    L.fr_plan_static_schedule.restype = C.c_int32
    L.fr_plan_pieces.restype = C.c_int64
    L.fr_plan_static_schedule.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.fr_plan_records.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.fr_plan_workspace_bytes.restype = C.c_int64
    L.fr_plan_workspace_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    L.fr_pipeline_create.restype = C.c_void_p
    L.fr_pipeline_destroy.restype = None
    L.fr_pipeline_destroy.argtypes = [C.c_void_p]
    L.fr_pipeline_info.restype = C.c_int64
    L.fr_pipeline_info.argtypes = [C.c_void_p, C.c_int32]
    L.fr_pipeline_workspace_bytes.restype = C.c_int64
    L.fr_pipeline_workspace_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.fr_plan_fits.restype = C.c_int32
    L.fr_plan_fits.argtypes = [C.c_void_p, C.c_int64]
    L.fr_plan_prepare.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
    L.fr_pipeline_prepare.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.fr_pipeline_compile_plan.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.fr_pipeline_prepare_cached.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.fr_pipeline_set_series_cuts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.fr_pipeline_set_argmax.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fr_select_ranks_begin.restype = C.c_void_p
    L.fr_select_ranks_end.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.fr_pipeline_set_preparation.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_double]
    _lib = L
    return L


def last_error() -> str:
    return lib().fr_last_error().decode()


def check(rc: int, what: str = "") -> None:
    if rc >= 0:
        return
    msg = last_error()
    if rc in (FR_E_DIM, FR_E_INDEX):
        raise IndexError(msg)
    if rc in (FR_E_ARG, FR_E_LIMIT):
        raise ValueError(msg)
    if rc == FR_E_NOMEM:
        raise MemoryError(msg)
    raise NativeError(f"{what}: {msg}" if what else msg)


def device_count() -> int:
    return int(lib().fr_device_count())


def require_device():
    """Returns the torch device to run on; raises when there is no GPU."""
    t = torch()
    if not t.cuda.is_available() or device_count() == 0:
        raise NativeError(
            "fruits_amd needs a HIP device (MI355X / gfx950); none is visible and "
            "there is deliberately no CPU fallback")
    return t.device("cuda", t.cuda.current_device())


_PREPARE_POOL = None


def _prepare_pool():
    """One helper thread for compilations nobody should wait for."""
    global _PREPARE_POOL
    if _PREPARE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _PREPARE_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fruits-prepare")
    return _PREPARE_POOL


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch().cuda.current_stream().cuda_stream)


def dptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def to_device(arr, dtype=None):
    """numpy (or torch) -> contiguous device tensor (float64 unless given)."""
    t = torch()
    dev = require_device()
    if isinstance(arr, t.Tensor):
        x = arr.to(device=dev)
        if dtype is not None:
            x = x.to(dtype)
        return x.contiguous()
    a = np.ascontiguousarray(arr, dtype=dtype or np.float64)
    # The copy from pageable memory is synchronous with the stream it is queued on - on the
    # caller's stream it would wait for everything queued there (Fruit.fit: the selection of the
    # slice before).  It goes through a stream of its own; the caller's stream waits for it on
    # the device.
    cur = t.cuda.current_stream(dev)
    if t.cuda.is_current_stream_capturing():
        return t.from_numpy(a).to(dev)
    up = _upload_stream(dev)
    with t.cuda.stream(up):
        x = t.from_numpy(a).to(dev)
    cur.wait_stream(up)
    x.record_stream(cur)
    return x


_upload_streams: dict = {}


def _upload_stream(dev):
    key = str(dev)
    st = _upload_streams.get(key)
    if st is None:
        st = _upload_streams[key] = torch().cuda.Stream(device=dev)
    return st


# Downloads of at least this many bytes land in page-locked memory from torch's caching host
# allocator (the array keeps its block; a freed one is reused): 72 MB of features come down at the
# link's rate instead of through the runtime's bounce buffers.  FRUITS_AMD_PINNED_MB = 0: never.
_PINNED_FROM = 1 << 20


def to_host(x) -> np.ndarray:
    x = x.detach()
    limit = int(os.environ.get("FRUITS_AMD_PINNED_MB", "4096")) << 20
    nbytes = x.numel() * x.element_size()
    if x.is_cuda and _PINNED_FROM <= nbytes <= limit:
        t = torch()
        try:
            h = t.empty(x.shape, dtype=x.dtype, pin_memory=True)
        except RuntimeError:       # (no page-locked memory left: the pageable path)
            return x.cpu().numpy()
        h.copy_(x.contiguous(), non_blocking=True)
        t.cuda.current_stream(x.device).synchronize()
        return h.numpy()
    return x.cpu().numpy()


# ----------------------------------------------------------------------- plan
class Plan:
    """Compiled device program of a word list (fr_plan_create)."""

    def __init__(self, words: Sequence[np.ndarray], depths: Sequence[int],
                 alphas: Optional[Sequence[np.ndarray]] = None, weighting: int = FR_W_NONE,
                 share_prefixes: bool = True, arctic: bool = False, bayesian: bool = False,
                 letter_sum: bool = False):
        L = lib()
        self._h = None
        mats = [np.ascontiguousarray(w, dtype=np.int32) for w in words]
        for m in mats:
            if m.ndim != 2:
                raise ValueError("word tables must be (L, Dw) int32")
        W = len(mats)
        exps = (np.concatenate([m.ravel() for m in mats]) if W else
                np.zeros(0, np.int32)).astype(np.int32)
        Ls = np.array([m.shape[0] for m in mats], dtype=np.int32)
        Dws = np.array([m.shape[1] for m in mats], dtype=np.int32)
        dep = np.asarray(depths, dtype=np.int32)
        if dep.shape != (W,):
            raise ValueError("one depth per word")
        al = None
        if weighting != FR_W_NONE:
            if alphas is None:
                raise ValueError("weighted plan needs alphas")
            al = np.concatenate([np.asarray(a, dtype=np.float32).ravel()
                                 for a in alphas]).astype(np.float32)
            if al.size != int(Ls.sum()):
                raise ValueError("Size of alpha array does not match word length")
        ip = C.POINTER(C.c_int32)
        h = L.fr_plan_create(
            C.c_int32(W), exps.ctypes.data_as(ip), Ls.ctypes.data_as(ip),
            Dws.ctypes.data_as(ip),
            al.ctypes.data_as(C.POINTER(C.c_float)) if al is not None else None,
            dep.ctypes.data_as(ip), C.c_int32(weighting),
            C.c_int32((1 if share_prefixes else 0) | (2 if arctic else 0)
                      | (4 if bayesian else 0) | (8 if letter_sum else 0)))
        if not h:
            raise ValueError(last_error())
        self._h = C.c_void_p(h)
        self.weighting = weighting
        self.n_words = W

    def __del__(self):
        try:
            if self._h is not None and _lib is not None:
                _lib.fr_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def info(self, what: int) -> int:
        return int(lib().fr_plan_info(self._h, C.c_int32(what)))

    @property
    def rows(self) -> int:
        return self.info(FR_INFO_ROWS)

    @property
    def nodes(self) -> int:
        return self.info(FR_INFO_NODES)

    @property
    def max_dim(self) -> int:
        return self.info(FR_INFO_MAX_DIM)

    @property
    def staged_rows(self) -> int:
        return self.info(FR_INFO_STAGED_ROWS)

    def fits(self, T: int) -> bool:
        """Whether a workgroup can stage the plan's rows (input dimensions + exp
        tables) of one time chunk in LDS (fr_plan_fits: the library's own rule)."""
        rc = int(lib().fr_plan_fits(self._h, int(T)))
        check(rc, "fr_plan_fits")
        return rc == 1

    def prepare(self, N: int, T: int, groups: int = 0) -> None:
        """One-time upload of the device tables for (N, T) batches (fr_plan_prepare):
        afterwards ``run`` only enqueues work and may be captured into a hipGraph."""
        check(lib().fr_plan_prepare(self._h, int(N), int(T), int(groups)), "fr_plan_prepare")

    @property
    def dims_used(self) -> int:
        return self.info(FR_INFO_DIMS_USED)

    def dump(self) -> np.ndarray:
        n = self.nodes
        buf = np.zeros((max(n, 1), 8), dtype=np.int32)
        got = lib().fr_plan_dump(self._h, buf.ctypes.data_as(C.POINTER(C.c_int32)),
                                 C.c_int32(buf.size))
        return buf[:got]

    def records(self, groups: int = 1) -> np.ndarray:
        """The device program for ``groups`` groups per series: (records, 16) int32."""
        n = int(lib().fr_plan_records(self._h, groups, None, 0))
        buf = np.zeros((max(n, 1), 16), dtype=np.int32)
        check(lib().fr_plan_records(self._h, groups, buf.ctypes.data, buf.size))
        return buf[:n]

    def jit(self, groups: int = 1, compile_only: bool = False):
        """fr_plan_jit: (count or code size, message)."""
        buf = C.create_string_buffer(8192)
        rc = lib().fr_plan_jit(self._h, int(groups), 1 if compile_only else 0, buf, len(buf))
        check(rc, "fr_plan_jit")
        return int(rc), buf.value.decode(errors="replace")

    def static_program_index(self, groups: int = 1) -> int:
        """1 + index of the pre-compiled static program this plan's records equal (0: none)."""
        return int(lib().fr_plan_info(self._h, FR_INFO_AOT_PROGRAM))

    def jit_loaded(self) -> int:
        """Number of run-time compiled static programs this plan holds on the device."""
        return int(lib().fr_plan_info(self._h, FR_INFO_JIT_PROGRAMS))

    def pieces(self, max_piece: int = 0):
        """The plan in pieces (fr_plan_pieces; csrc/plan.h, PiecedProgram) as a dict, or None when
        the plan has no such cover: ``types`` - per piece type its body / chain records
        ((records, 16) int32), items ((items, 4): chain byte offset, walk position of the body's
        first row, nodes of the unit in front), ``unit_begin``, ``unit_row0`` - and
        ``row_of_walk``, the output row at every walk position."""
        L = lib()
        n = int(L.fr_plan_pieces(self._h, C.c_int32(max_piece), None, C.c_int64(0)))
        if n <= 0:
            return None
        buf = np.zeros(n, dtype=np.int32)
        L.fr_plan_pieces(self._h, C.c_int32(max_piece), buf.ctypes.data_as(C.POINTER(C.c_int32)),
                         C.c_int64(n))
        n_types, K, chain_nodes, nodes = (int(v) for v in buf[:4])
        at, types = 4, []
        for _ in range(n_types):
            (body_nodes, body_rows, levels, units, n_items, max_unit_nodes, n_recs,
             max_unit_rows) = (int(v) for v in buf[at:at + 8])
            at += 8
            recs = buf[at:at + 16 * n_recs].reshape(n_recs, 16).copy()
            at += 16 * n_recs
            items = buf[at:at + 4 * n_items].reshape(n_items, 4).copy()
            at += 4 * n_items
            unit_begin = buf[at:at + units + 1].copy()
            at += units + 1
            unit_row0 = buf[at:at + units].copy()
            at += units
            types.append(dict(body_nodes=body_nodes, body_rows=body_rows, levels=levels, units=units,
                              max_unit_nodes=max_unit_nodes, max_unit_rows=max_unit_rows, recs=recs,
                              items=items, unit_begin=unit_begin, unit_row0=unit_row0))
        return dict(types=types, K=K, chain_nodes=chain_nodes, nodes=nodes,
                    row_of_walk=buf[at:at + K].copy())

    def static_schedule(self, groups: int = 1):
        """(header dict, (entries, 16) int32) of the plan's static schedule, or None."""
        n = int(lib().fr_plan_static_schedule(self._h, groups, None, 0))
        if n <= 0:
            return None
        buf = np.zeros(32 + n * 16, dtype=np.int32)
        check(lib().fr_plan_static_schedule(self._h, groups, buf.ctypes.data, buf.size))
        rows, G = int(buf[1]), int(buf[3])
        head = {"entries": n, "rows": rows, "frames": int(buf[2]), "groups": G,
                "row_src": [int(v) for v in buf[4:4 + rows]],
                "group_begin": [int(v) for v in buf[8:8 + G]],
                "group_rows": [int(v) for v in buf[16:16 + G]]}
        return head, buf[32:].reshape(n, 16)

    def workspace_bytes(self, N: int, T: int, lookup_rows: int) -> int:
        return int(lib().fr_plan_workspace_bytes(self._h, N, T, lookup_rows))

    def run(self, Xd, lookup_d=None, out=None, layout: str = "KNT", groups: int = 0,
            work=None, strides=None):
        """Launches the trie walk on the current stream.

        Xd (N,D,T) f64 cuda; lookup_d (1|N, T) f64 cuda or None.
        layout "KNT" -> (K,N,T) (fruits/iss/iss.py:46), "NKT" -> (N,K,T)
        (fruits/iss/semiring.py:177)."""
        t = torch()
        if Xd.dtype != t.float64 or Xd.dim() != 3 or not Xd.is_contiguous():
            raise TypeError("X must be a contiguous float64 (N, D, T) device tensor")
        N, D, T = Xd.shape
        K = self.rows
        if out is None:
            shape = (K, N, T) if layout == "KNT" else (N, K, T)
            out = t.empty(shape, dtype=t.float64, device=Xd.device)
        if strides is not None:
            sk, sn = strides   # caller-laid-out buffer (element strides of k and n)
        elif layout == "KNT":
            sk, sn = N * T, T
        else:
            sk, sn = T, K * T
        rows = 0
        if self.weighting != FR_W_NONE:
            if lookup_d is None:
                raise ValueError("weighted plan needs a lookup")
            rows = int(lookup_d.shape[0])
            if lookup_d.shape[-1] != T or not lookup_d.is_contiguous():
                raise ValueError("lookup must be contiguous (1|N, T)")
        wb = self.workspace_bytes(N, T, rows)
        if wb > 0 and (work is None or work.numel() < wb):
            work = t.empty(wb, dtype=t.uint8, device=Xd.device)
        rc = lib().fr_iss_run(
            self._h, dptr(Xd), C.c_int64(N), C.c_int64(D), C.c_int64(T),
            dptr(lookup_d if self.weighting != FR_W_NONE else None), C.c_int64(rows),
            dptr(out), C.c_int64(sk), C.c_int64(sn), dptr(work),
            C.c_int64(work.numel() if work is not None else 0), C.c_int32(groups),
            stream_ptr())
        check(rc, "fr_iss_run")
        return out


class CosPlan(Plan):
    """Device program of a cosine weighted ISS (fr_plan_create_coswiss): rows
    word-major, ``len(freqs)`` rows per word."""

    MAX_EXPONENT = 8
    MAX_LETTERS = 16

    def __init__(self, words: Sequence[np.ndarray], freqs, exponent: int, total: bool):
        L = lib()
        self._h = None
        mats = [np.ascontiguousarray(w, dtype=np.int32) for w in words]
        W = len(mats)
        exps = (np.concatenate([m.ravel() for m in mats]) if W else
                np.zeros(0, np.int32)).astype(np.int32)
        Ls = np.array([m.shape[0] for m in mats], dtype=np.int32)
        Dws = np.array([m.shape[1] for m in mats], dtype=np.int32)
        fq = np.asarray(freqs, dtype=np.float32).ravel()
        ip = C.POINTER(C.c_int32)
        h = L.fr_plan_create_coswiss(
            C.c_int32(W), exps.ctypes.data_as(ip), Ls.ctypes.data_as(ip), Dws.ctypes.data_as(ip),
            C.c_int32(fq.size), fq.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(exponent),
            C.c_int32(1 if total else 0))
        if not h:
            raise ValueError(last_error())
        self._h = C.c_void_p(h)
        self.weighting = FR_W_NONE
        self.n_words = W


def coswiss_set_dropout(plan: "CosPlan", indices, T: int) -> None:
    """fr_coswiss_set_dropout: ``indices`` (W, F, Lmax, rate) int32 as drawn by CosWISS._fit;
    ``None`` switches the mask off."""
    if indices is None:
        check(lib().fr_coswiss_set_dropout(plan._h, None, C.c_int32(0), C.c_int32(0),
                                           C.c_int64(max(int(T), 1))), "fr_coswiss_set_dropout")
        return
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    check(lib().fr_coswiss_set_dropout(plan._h, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                       C.c_int32(idx.shape[2]), C.c_int32(idx.shape[3]),
                                       C.c_int64(int(T))), "fr_coswiss_set_dropout")


def coswiss_ffn(Xd, A, b, Cm, out):
    """fr_coswiss_ffn: out (N, D, T) = C relu(A x + b) per time step; A (hidden, D), b
    (hidden), Cm (D, hidden) host arrays."""
    N, D, T = (int(v) for v in Xd.shape)
    Ad = to_device(np.ascontiguousarray(A, dtype=np.float64))
    bd = to_device(np.ascontiguousarray(b, dtype=np.float64))
    Cd = to_device(np.ascontiguousarray(Cm, dtype=np.float64))
    check(lib().fr_coswiss_ffn(dptr(Xd), C.c_int64(N), C.c_int64(D), C.c_int64(T), dptr(Ad),
                               dptr(bd), dptr(Cd), C.c_int32(int(A.shape[0])), dptr(out),
                               stream_ptr()), "fr_coswiss_ffn")
    return out


class Pipeline:
    """ISS + sieves fused into one launch (fr_pipeline_*): the (K, N, T) tensor is
    never written.  ``sieves`` is a list of ``(kind, inc, cut_row, Q1)`` with
    ``cut_row`` the transformed int64 cuts (sorted, leading 0) for length ``T``.
    Raises ValueError when a sieve is outside the fused set."""

    def __init__(self, plan: Plan, sieves, T: int, argmax_lengths=None):
        """``argmax_lengths``: the plan holds every prefix of words of these lengths and the
        pipeline's rows are those of Arctic(argmax=True) (fr_pipeline_set_argmax)."""
        L = lib()
        self._h = None
        self.plan = plan
        self.T = int(T)
        n = len(sieves)
        kinds = np.array([s[0] for s in sieves], dtype=np.int32)
        incs = np.array([s[1] for s in sieves], dtype=np.int32)
        C1 = np.array([len(s[2]) for s in sieves], dtype=np.int32)
        Q1 = np.array([s[3] for s in sieves], dtype=np.int32)
        cuts = np.concatenate([np.asarray(s[2], dtype=np.int64) for s in sieves]).astype(np.int64)
        ip = C.POINTER(C.c_int32)
        h = L.fr_pipeline_create(plan._h, C.c_int32(n), kinds.ctypes.data_as(ip),
                                 incs.ctypes.data_as(ip), C1.ctypes.data_as(ip),
                                 Q1.ctypes.data_as(ip),
                                 cuts.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(self.T))
        if not h:
            msg = last_error()
            raise (IndexError if "out of bounds" in msg else ValueError)(msg)
        self._h = C.c_void_p(h)
        self.raw_dims = 0    # > 0: fused preparation, run() takes the raw input
        self.rows = plan.rows
        if argmax_lengths is not None:
            lens = np.ascontiguousarray(argmax_lengths, dtype=np.int32)
            rc = L.fr_pipeline_set_argmax(self._h, C.c_int32(len(lens)), lens.ctypes.data_as(ip))
            if rc != 0:
                msg = last_error()
                L.fr_pipeline_destroy(self._h)
                self._h = None
                raise ValueError(msg)
            self.rows = int(sum(int(n) + int(n) * (int(n) + 1) // 2 for n in lens))
        self.per_sum = int(L.fr_pipeline_info(self._h, 0))
        self.q_stride = int(L.fr_pipeline_info(self._h, 1))
        self.n_features = int(L.fr_pipeline_info(self._h, 2))

    def __del__(self):
        try:
            if self._h is not None and _lib is not None:
                _lib.fr_pipeline_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_quantiles(self, quant: np.ndarray) -> None:
        """``quant`` (K, q_stride) host array: the sorted thresholds of every
        iterated sum's band sieves, in sieve order."""
        q = np.ascontiguousarray(quant, dtype=np.float64)
        if q.shape != (self.rows, self.q_stride):
            raise ValueError("quantile table must be (K, q_stride)")
        check(lib().fr_pipeline_set_quantiles(self._h, q.ctypes.data_as(C.POINTER(C.c_double))),
              "fr_pipeline_set_quantiles")

    def set_series_cuts(self, table) -> None:
        """``table`` (N, slots) int32 device tensor: the per-series boundaries of the sieves
        created with FR_SIEVE_SERIES_CUTS, for the following runs on exactly N series."""
        t = torch()
        if table.dtype != t.int32 or table.dim() != 2 or not table.is_contiguous():
            raise TypeError("the cut table must be a contiguous int32 (N, slots) device tensor")
        check(lib().fr_pipeline_set_series_cuts(self._h, C.c_void_p(table.data_ptr()),
                                                C.c_int64(table.shape[0]), C.c_int32(table.shape[1])),
              "fr_pipeline_set_series_cuts")
        self._series_cuts = table    # (keeps the tensor alive while the pipeline points at it)

    def set_preparation(self, D: int, inc_lag: int = 0, as_new: bool = False,
                        standardize: int = 0, std_eps: float = 1e-5) -> bool:
        """fr_pipeline_set_preparation: ``run`` then takes the RAW (N, D, T) input and forms
        INC / NEW(INC) / STD rows while staging.  False when this plan's kernel has no fused
        staging (the caller keeps materialising the prepared input)."""
        rc = lib().fr_pipeline_set_preparation(self._h, int(D), int(inc_lag), 1 if as_new else 0,
                                               int(standardize), float(std_eps))
        if rc == FR_E_LIMIT:
            self.raw_dims = 0
            return False
        check(rc, "fr_pipeline_set_preparation")
        self.raw_dims = int(D) if (inc_lag or standardize) else 0
        return self.raw_dims > 0

    # plans up to this size are compiled as straight-line code by a prepare nobody waits for, and
    # the node shapes of plans beyond 128 nodes (a few seconds of compiler either way; the
    # interpreter joins its helper threads at exit) - between, ten seconds and more
    QUICK_PLAN_NODES = 48

    def prepare(self, N: int, groups: int = 0, plan_too: bool = True) -> None:
        """fr_pipeline_prepare: uploads the plan's tables for batches of N series so
        that ``run`` only enqueues work (hipGraph capture), and compiles the pipeline's own
        kernel - the fused walk with its sieves as compile-time constants (hipRTC, cached on disk;
        FRUITS_HIP_JIT=0: not); ``plan_too``: also the variant that knows the plan
        (fr_pipeline_compile_plan: a plan of at most 128 nodes as straight-line code, of a larger
        one the node shapes - seconds to tens of seconds the first time on a machine)."""
        check(lib().fr_pipeline_prepare(self._h, int(N), int(groups)), "fr_pipeline_prepare")
        if plan_too:
            check(lib().fr_pipeline_compile_plan(self._h, int(N), int(groups)), "fr_pipeline_compile_plan")

    def prepare_cached(self, N: int, groups: int = 0) -> int:
        """fr_pipeline_prepare_cached: the uploads, and the pipeline's own kernels if an earlier
        process on this machine compiled them (the disk cache: milliseconds; nothing is
        compiled).  Returns the number of kernels the pipeline now holds."""
        check(lib().fr_pipeline_prepare_cached(self._h, int(N), int(groups)), "fr_pipeline_prepare_cached")
        return self.jit_loaded()

    def prepare_in_background(self, N: int, groups: int = 0):
        """``prepare`` on a helper thread: the caller goes on with the generic kernel and a later
        ``run`` picks the pipeline's own kernel up once it is compiled and loaded (the library
        guards the pipeline's compiled kernels with a lock).  Returns the future; a failure to
        compile is not an error - the generic kernel stays - anything else is re-raised by
        ``future.result()``."""
        device = torch().cuda.current_device()

        def work():
            torch().cuda.set_device(device)      # (the current device is per thread)
            # (self: the pipeline outlives the compilation)
            nodes = self.plan.nodes
            self.prepare(N, groups, plan_too=nodes <= self.QUICK_PLAN_NODES or nodes > 128)
        self._pending = _prepare_pool().submit(work)

        def report(fut):     # (nobody may ever call result(): a failure is said once, not swallowed)
            err = fut.exception()
            if err is not None:
                import warnings
                warnings.warn(f"fruits_amd: compiling a pipeline's own kernels failed ({err}); "
                              "the generic kernel keeps running it", RuntimeWarning)
        self._pending.add_done_callback(report)
        return self._pending

    def jit_loaded(self, static_only: bool = False) -> int:
        """Run-time compiled kernels this pipeline holds (``static_only``: those with the plan
        as straight-line code)."""
        return int(lib().fr_pipeline_info(self._h, 4 if static_only else 3))

    def bundle(self, quant: np.ndarray, out_dir: str, groups: int = 1) -> int:
        """fr_pipeline_bundle: compiles this pipeline's kernels (the sieves as immediates; the
        plan as straight-line code or in pieces) into ``out_dir`` without a device; ``quant``
        (K, q_stride) carries the infinities of the thresholds (finite values are placeholders)."""
        q = np.ascontiguousarray(quant, dtype=np.float64)
        if q.shape != (self.plan.rows, self.q_stride):
            raise ValueError("quantile table must be (K, q_stride)")
        buf = C.create_string_buffer(4096)
        L = lib()
        L.fr_pipeline_bundle.restype = C.c_int32
        rc = L.fr_pipeline_bundle(self._h, q.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(groups),
                                  os.fsencode(out_dir), buf, C.c_int64(len(buf)))
        if rc < 0:
            raise NativeError(f"fr_pipeline_bundle: {last_error()}")
        return int(rc)

    def fully_compiled(self) -> bool:
        """Whether the pipeline already holds the kernels a ``prepare`` would give it: the plan's
        own (in pieces, or as straight-line code) - nothing left for a compiler to do."""
        return self.pieces_loaded() > 0 or self.jit_loaded(static_only=True) > 0

    def pieces_loaded(self) -> int:
        """Kernels of piece types this pipeline holds (a large plan in pieces)."""
        return int(lib().fr_pipeline_info(self._h, 5))

    def run(self, Xd, lookup_d, feats=None, groups: int = 0, work=None):
        t = torch()
        if Xd.dtype != t.float64 or Xd.dim() != 3 or not Xd.is_contiguous():
            raise TypeError("X must be a contiguous float64 (N, D, T) device tensor")
        N, D, T = Xd.shape
        if feats is None:
            feats = t.empty((N, self.n_features), dtype=t.float64, device=Xd.device)
        rows = 0
        if self.plan.weighting != FR_W_NONE:
            if lookup_d is None:
                raise ValueError("weighted plan needs a lookup")
            rows = int(lookup_d.shape[0])
        wb = int(lib().fr_pipeline_workspace_bytes(self._h, N, rows))
        if wb > 0 and (work is None or work.numel() < wb):
            work = t.empty(wb, dtype=t.uint8, device=Xd.device)
        rc = lib().fr_pipeline_run(
            self._h, dptr(Xd), C.c_int64(N), C.c_int64(D), C.c_int64(T),
            dptr(lookup_d if self.plan.weighting != FR_W_NONE else None), C.c_int64(rows),
            dptr(feats), C.c_int64(feats.stride(0)), dptr(work),
            C.c_int64(work.numel() if work is not None else 0), C.c_int32(groups), stream_ptr())
        check(rc, "fr_pipeline_run")
        return feats


# ----------------------------------------------------------------------- kernels
def iterated_sum_fast_host(Z, word, alpha, lookup, extended, total_weighting, arctic=False,
                           bayesian=False):
    """fr_iterated_sum_fast_host: host arrays in / out (the literal drop-in of
    Semiring.iterated_sum_fast, fruits/iss/semiring.py:43-52)."""
    require_device()
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    word = np.ascontiguousarray(word, dtype=np.int32)
    N, D, T = Z.shape
    Lw, Dw = word.shape
    out = np.zeros((N, int(extended), T))
    al = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.float32)
    lk = None if lookup is None else np.ascontiguousarray(lookup, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    rc = lib().fr_iterated_sum_fast_host(
        Z.ctypes.data_as(dp), C.c_int64(N), C.c_int64(D), C.c_int64(T),
        word.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(Lw), C.c_int32(Dw),
        al.ctypes.data_as(C.POINTER(C.c_float)) if al is not None else None,
        lk.ctypes.data_as(dp) if lk is not None else None, C.c_int64(int(extended)),
        C.c_int32((1 if total_weighting else 0) | (2 if arctic else 0) | (4 if bayesian else 0)),
        out.ctypes.data_as(dp))
    check(rc, "fr_iterated_sum_fast_host")
    return out


def increments(Xd, shift: int, head_src=None, head: int = 0, out=None):
    t = torch()
    if out is None:
        out = t.empty_like(Xd)
    T = Xd.shape[-1]
    rows = Xd.numel() // T if T else 0
    rc = lib().fr_increments(dptr(Xd), C.c_int64(rows), C.c_int64(T), C.c_int64(shift),
                             dptr(out), dptr(head_src), C.c_int64(head), stream_ptr())
    check(rc, "fr_increments")
    return out


FR_LOOKUP_FAST = 16


def pathlen_lookup(Xd, norm: int = 1, relative: int = 0, scale: float = 50.0,
                   exact: bool = True):
    """relative: 0 plain, 1 divide by (last + 1e-5) first, 2 = raw cumulative
    path length without normalisation (the SharedSeedCache entry)."""
    t = torch()
    N, D, T = Xd.shape
    out = t.empty((N, T), dtype=t.float64, device=Xd.device)
    rc = lib().fr_pathlen_lookup(dptr(Xd), C.c_int64(N), C.c_int64(D), C.c_int64(T),
                                 C.c_int32(norm | (0 if exact else FR_LOOKUP_FAST)),
                                 C.c_int32(relative), C.c_double(scale), dptr(out), stream_ptr())
    check(rc, "fr_pathlen_lookup")
    return out


def sieve(kind: int, Ad, inc: int, cuts_d, q_d, out, out_col: int = 0):
    """Writes the features of one sieve on a (N, T) device array into columns
    [out_col, out_col + nfeatures) of the (N, F) device tensor ``out``."""
    N, T = Ad.shape
    C1 = int(cuts_d.shape[1])
    Q1 = int(q_d.numel()) if q_d is not None else 0
    base = out.data_ptr() + 8 * out_col
    rc = lib().fr_sieve(C.c_int32(kind), dptr(Ad), C.c_int64(N), C.c_int64(T),
                        C.c_int64(Ad.stride(0)), C.c_int32(inc), dptr(cuts_d),
                        C.c_int64(cuts_d.shape[0]), C.c_int32(C1), dptr(q_d), C.c_int32(Q1),
                        C.c_void_p(base), C.c_int64(out.stride(0)), stream_ptr())
    check(rc, "fr_sieve")
    return out


def pre_transform(Ad, inc: int):
    """IncrementSieve._pre_transform for inc >= 0 on a (N, T) device array."""
    t = torch()
    out = t.empty_like(Ad)
    N, T = Ad.shape
    rc = lib().fr_pre_transform(dptr(Ad), C.c_int64(N), C.c_int64(T), C.c_int64(Ad.stride(0)),
                                C.c_int32(inc), dptr(out), stream_ptr())
    check(rc, "fr_pre_transform")
    return out


def select_ranks(block, job_row, job_inc, job_rank, stream=None) -> np.ndarray:
    """Order statistics of differenced (N, T) blocks of the (rows, N, T) device
    tensor ``block`` (fr_select_ranks); exact.  ``stream``: the HIP stream ``block`` was written
    on, when the call comes from another thread than the one that wrote it (the current device
    and stream are per thread)."""
    if stream is not None:
        torch().cuda.set_device(block.device)
    rows, N, T = block.shape
    jr = np.ascontiguousarray(job_row, dtype=np.int32)
    ji = np.ascontiguousarray(job_inc, dtype=np.int32)
    jk = np.ascontiguousarray(job_rank, dtype=np.int64)
    out = np.zeros(len(jr))
    if not block.is_contiguous():
        raise ValueError("block must be contiguous")
    rc = lib().fr_select_ranks(
        dptr(block), C.c_int64(rows), C.c_int64(N), C.c_int64(T), C.c_int32(len(jr)),
        jr.ctypes.data_as(C.POINTER(C.c_int32)), ji.ctypes.data_as(C.POINTER(C.c_int32)),
        jk.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_double)),
        stream_ptr() if stream is None else stream)
    check(rc, "fr_select_ranks")
    return out


class Selection:
    """fr_select_ranks_begin / _end: the selection is queued on the current stream and the call
    returns; ``result()`` waits for it and returns the values (once)."""

    def __init__(self, block, job_row, job_inc, job_rank):
        rows, N, T = block.shape
        if not block.is_contiguous():
            raise ValueError("block must be contiguous")
        jr = np.ascontiguousarray(job_row, dtype=np.int32)
        ji = np.ascontiguousarray(job_inc, dtype=np.int32)
        jk = np.ascontiguousarray(job_rank, dtype=np.int64)
        self._n = len(jr)
        self._block = block      # (read by the passes until result())
        self._vals = None
        L = lib()
        h = L.fr_select_ranks_begin(
            dptr(block), C.c_int64(rows), C.c_int64(N), C.c_int64(T), C.c_int32(self._n),
            jr.ctypes.data_as(C.POINTER(C.c_int32)), ji.ctypes.data_as(C.POINTER(C.c_int32)),
            jk.ctypes.data_as(C.POINTER(C.c_int64)), stream_ptr())
        if not h:
            raise NativeError(f"fr_select_ranks_begin: {last_error()}")
        self._h = C.c_void_p(h)

    def result(self) -> np.ndarray:
        if self._vals is None:
            out = np.zeros(self._n)
            h, self._h = self._h, None
            check(lib().fr_select_ranks_end(h, out.ctypes.data_as(C.POINTER(C.c_double))),
                  "fr_select_ranks_end")
            self._vals, self._block = out, None
        return self._vals

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and _lib is not None:
                _lib.fr_select_ranks_end(self._h, None)     # (waits, frees the handle and its scratch)
                self._h = None
        except Exception:
            pass


def standardize(Xd, div_std: bool, eps: float):
    t = torch()
    out = t.empty_like(Xd)
    T = Xd.shape[-1]
    rows = Xd.numel() // T if T else 0
    rc = lib().fr_standardize(dptr(Xd), C.c_int64(rows), C.c_int64(T),
                              C.c_int32(1 if div_std else 0), C.c_double(eps), dptr(out),
                              stream_ptr())
    check(rc, "fr_standardize")
    return out


def coswiss_combine(terms_d, begin_d, coeff_d, desc_d, trig_d, out, out_row_stride: int):
    """out[j] = sum of the weighted terms of output row j (fr_coswiss_combine);
    ``out`` is a device view whose rows are ``out_row_stride`` doubles apart."""
    n_terms, N, T = (int(v) for v in terms_d.shape)
    n_out = int(begin_d.shape[0]) - 1
    rc = lib().fr_coswiss_combine(dptr(terms_d), C.c_int64(n_terms), C.c_int64(N), C.c_int64(T),
                                  C.c_int32(n_out), dptr(begin_d), dptr(coeff_d), dptr(desc_d),
                                  dptr(trig_d), dptr(out), C.c_int64(out_row_stride),
                                  stream_ptr())
    check(rc, "fr_coswiss_combine")
    return out


def release_scratch() -> None:
    """Frees the grow-only device scratch of fr_select_ranks (kept between the calls of one fit)."""
    check(lib().fr_release_scratch(), "fr_release_scratch")


def nan_to_num(xd):
    """In place np.nan_to_num(x, nan=0.0) of a contiguous float64 device tensor."""
    if not xd.is_contiguous():
        raise TypeError("nan_to_num needs a contiguous tensor")
    check(lib().fr_nan_to_num(dptr(xd), C.c_int64(xd.numel()), stream_ptr()), "fr_nan_to_num")
    return xd


def arctic_argmax(Vd, word_lengths) -> "object":
    """fr_arctic_argmax: the rows of Arctic(argmax=True) from the running maxima ``Vd``
    (sum(L), N, T) of all prefixes of all words (lengths ``word_lengths``)."""
    t = torch()
    rows, N, T = (int(v) for v in Vd.shape)
    jobs, v0, o0 = [], 0, 0
    for L in word_lengths:
        for k in range(L):
            jobs.append((v0, k, o0 + k + k * (k + 1) // 2))
        v0 += L
        o0 += L + L * (L + 1) // 2
    if v0 != rows:
        raise ValueError("Vd must hold one row per prefix of every word")
    if max(word_lengths, default=0) > 63:
        raise NotImplementedError("Arctic argmax: words of more than 63 letters")
    out = t.empty((o0, N, T), dtype=t.float64, device=Vd.device)
    if not jobs or N == 0 or T == 0:
        return out
    jd = to_device(np.asarray(jobs, dtype=np.int32), dtype=np.int32)
    P = t.empty_like(Vd)
    check(lib().fr_arctic_argmax(dptr(Vd), C.c_int64(rows), C.c_int64(N), C.c_int64(T),
                                 C.c_int32(len(jobs)), dptr(jd), dptr(P), dptr(out), stream_ptr()),
          "fr_arctic_argmax")
    return out
