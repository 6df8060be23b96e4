"""Pipeline orchestration: ``Fruit`` and ``FruitSlice`` (mirrors fruits/fruit.py).

The public behaviour is the reference's - slices of preparateurs -> ISS ->
sieves, per-iterated-sum fitted sieve copies, feature column order (slice, then
iterated sum, then sieve, then the sieve's own (segment, band) order) and the
final ``nan_to_num``.  The execution is not: the batch is uploaded to HBM once,
every stage runs as HIP kernels on device tensors, iterated sums are produced
by one launch per word batch instead of one numba call per word, sieves write
straight into the ``(N, F)`` feature tensor, and only that tensor travels back.
"""
from __future__ import annotations

import inspect
import os
from typing import Callable, Generator, Literal, Optional, Union

import numpy as np

from . import _native as nat
from .cache import SharedSeedCache
from .callback import AbstractCallback
from .iss.iss import ISS
from .preparation.abstract import Preparateur
from .seed import Seed
from .sieving.abstract import FeatureSieve


def _check_batch(X) -> np.ndarray:
    if not isinstance(X, np.ndarray) or X.ndim != 3:
        raise TypeError("input has to be an array of shape (N, D, T)")
    if X.dtype != np.float64:
        raise TypeError("input has to be float64 (the reference's numba kernels "
                        "accept nothing else)")
    return X


_SELECTIONS_IN_FLIGHT = 6     # of the 8 the library allows per device (fr_select_ranks_begin)


class _FittedRows:
    """The fitted sieve copies of a slice, one list per iterated sum - what fruits/fruit.py:462-476
    builds as it fits - held as ARRAYS: per sieve the thresholds of all rows, computed from the
    device's order statistics with numpy (np.quantile's interpolation over the rows at once).  The
    copy objects are made when somebody asks for a row; the fused pipeline reads the arrays.
    (fruit_reduced: 633 rows x 7 sieves - as eager Python objects, each fitted by a call of its
    own, they were 20 of the 36 ms of Fruit.fit, all on the thread that also feeds the device.)"""

    def __init__(self, sieves, cache) -> None:
        self._sieves = list(sieves)
        self._cache = cache
        self._segments: list = []     # [first row, rows, per sieve (rows, len(q)) array or None]
        self._n = 0
        self._made: dict = {}

    def __getstate__(self):
        # (copies somebody changed are part of the state; the cache is the process's)
        return {"_sieves": self._sieves, "_cache": None, "_n": self._n, "_made": {},
                "_segments": [[r0, n, [None if q is None else q.copy() for q in qs]]
                              for r0, n, qs in self._with_made()]}

    def _with_made(self):
        for r0, n, qs in self._segments:
            qs = [None if q is None else q.copy() for q in qs]
            for k, fitted in self._made.items():
                if r0 <= k < r0 + n:
                    for i, sv in enumerate(fitted):
                        if qs[i] is not None:
                            qs[i][k - r0] = sv._quantiles
            yield r0, n, qs

    def add_rows(self, n: int) -> list:
        """A run of ``n`` more rows; returns its (still empty) list of threshold arrays."""
        seg = [self._n, n, [None] * len(self._sieves)]
        self._segments.append(seg)
        self._n += n
        return seg[2]

    def __len__(self) -> int:
        return self._n

    def __bool__(self) -> bool:
        return self._n > 0

    def __iter__(self):
        return (self[k] for k in range(self._n))

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self._n))]
        if k < 0:
            k += self._n
        if not 0 <= k < self._n:
            raise IndexError(k)
        fitted = self._made.get(k)
        if fitted is None:
            r0, _, qs = next(seg for seg in self._segments if seg[0] <= k < seg[0] + seg[1])
            fitted = [sv.copy() for sv in self._sieves]
            for sv, q in zip(fitted, qs):
                sv._cache = self._cache
                if q is not None:
                    sv._quantiles = q[k - r0].copy()
            self._made[k] = fitted
        return fitted

    def thresholds(self, i: int, rows) -> Optional[np.ndarray]:
        """(len(rows), len(q)) thresholds of sieve ``i`` (None: it is not fitted on data)."""
        if not self._segments or self._segments[0][2][i] is None:
            return None
        if len(self._segments) == 1 and not self._made:
            table = self._segments[0][2][i]
        else:
            table = np.concatenate([qs[i] for _, _, qs in self._with_made()])
        return table[np.asarray(rows, dtype=np.int64)]


def _interpolated_quantiles(sieve, reqs, lo_vals: np.ndarray, hi_vals: np.ndarray) -> np.ndarray:
    """SegmentSieve._set_quantiles_from_stats for all rows at once: (rows, len(q)) thresholds
    from the (rows, len(reqs)) order statistics np.quantile interpolates between."""
    qs = np.empty((lo_vals.shape[0], len(sieve._q)))
    qs[:] = [np.inf if q == 1.0 else (-np.inf if q == -1.0 else 0.0) for q in sieve._q]
    for j, (i, _, _, gamma) in enumerate(reqs):
        # numpy's _lerp (lib/_function_base_impl.py): a + (b-a)*t, from the other end
        # when t >= 0.5
        a, b = lo_vals[:, j], hi_vals[:, j]
        d = b - a
        qs[:, i] = a + d * gamma if gamma < 0.5 else b - d * (1 - gamma)
    return np.sort(qs, axis=1)


class Fruit:
    """Feature extractor made of one or more :class:`FruitSlice` objects whose
    features are concatenated.

    .. code-block:: python

        fruit = fruits_amd.Fruit("My Fruit")
        fruit.add(fruits_amd.preparation.INC)
        fruit.add(fruits_amd.ISS(fruits_amd.words.of_weight(2, dim=3),
                                 mode=fruits_amd.ISSMode.EXTENDED))
        fruit.add(fruits_amd.sieving.NPI(q=(0.5, 1.0)), fruits_amd.sieving.END)
        fruit.fit(X)
        features = fruit.transform(X)
    """

    def __init__(self, name: str = "") -> None:
        self.name: str = name
        self._slices: list[FruitSlice] = []
        self._slc_index: int = 0
        self._fitted: bool = False
        self._cursor: int = -1

    # ---- structure ------------------------------------------------------------
    def cut(self, slice: Optional["FruitSlice"] = None) -> None:
        """Appends a (new, empty) slice and makes it the current one."""
        self._slices.append(FruitSlice() if slice is None else slice)
        self._slc_index = len(self._slices) - 1
        self._fitted = False

    def copycut(self) -> None:
        self.cut(self.get_slice().deepcopy())

    def get_slice(self, index: Optional[int] = None) -> "FruitSlice":
        return self._slices[self._slc_index if index is None else index]

    def switch_slice(self, index: int) -> None:
        if not (0 <= index < len(self._slices)):
            raise IndexError("Index has to be in [0, len(self)-1]")
        self._slc_index = index

    def add(self, *objects: Union[Seed, Callable[[], Seed]]) -> None:
        """Adds preparateurs, ISS or sieves to the current slice."""
        if not self._slices:
            self.cut()
        self._slices[self._slc_index].add(*objects)
        self._fitted = False

    def nfeatures(self) -> int:
        return sum(slc.nfeatures() for slc in self._slices)

    # ---- fit / transform ----------------------------------------------------
    def fit(self, X: np.ndarray, cache: Optional[SharedSeedCache] = None) -> None:
        X = _check_batch(X)
        cache_ = SharedSeedCache(X) if cache is None else cache
        # the slices are started one after the other; the device-side selections of a slice's
        # thresholds are only queued (fr_select_ranks_begin) and waited for at the end
        deferred: list = []
        try:
            for slc in self._slices:
                slc.fit(X, cache=cache_, deferred=deferred)
        finally:
            for finish in deferred:
                finish()
        # (the selection scratch of the device-side fit - a blob of device and page-locked memory
        # per slice in flight, some 50 MB each - lives outside torch's allocator and is kept for
        # the next fit: allocating it is a millisecond per blob on the fit's critical path.
        # fruits_amd.release_scratch() hands it back.)
        self._fitted = True

    def transform(self, X: np.ndarray,
                  callbacks: Optional[list[AbstractCallback]] = None,
                  cache: Optional[SharedSeedCache] = None) -> np.ndarray:
        """``(N, nfeatures)`` features of all slices.

        Raises:
            RuntimeError: if :meth:`fit` was not called.
        """
        callbacks = callbacks or []
        if not self._fitted:
            raise RuntimeError("Missing call of self.fit")
        X = _check_batch(X)
        cache_ = SharedSeedCache(X) if cache is None else cache
        t = nat.torch()
        blocks = []
        for slc in self._slices:
            for cb in callbacks:
                cb.on_next_slice()
            blocks.append(slc.transform_device(X, callbacks, cache_))
        # assembled on the device: one download instead of a strided host copy per slice
        result = blocks[0] if len(blocks) == 1 else t.cat(blocks, dim=1)
        return nat.to_host(nat.nan_to_num(result.contiguous()))

    def fit_transform(self, X: np.ndarray,
                      callbacks: Optional[list[AbstractCallback]] = None) -> np.ndarray:
        self.fit(X)
        return self.transform(X, callbacks=callbacks)

    # ---- the fitted state, to be handed to other processes (fruits_amd.parallel) ----
    def fit_state(self) -> bytes:
        """Everything ``fit`` derived from the data, pickled: the fitted seeds of every slice."""
        if not self._fitted:
            raise RuntimeError("Missing call of self.fit")
        import pickle
        return pickle.dumps([slc._fit_state() for slc in self._slices])

    def load_fit_state(self, state: bytes) -> None:
        """Makes this fruit (same configuration) the fitted fruit ``state`` was taken from."""
        import pickle
        states = pickle.loads(state)
        if len(states) != len(self._slices):
            raise ValueError("the state belongs to a fruit with another number of slices")
        for slc, st in zip(self._slices, states):
            slc._load_fit_state(st)
        self._fitted = True

    # ---- introspection ------------------------------------------------------
    def summary(self) -> str:
        bar = 80 * "="
        ident = "Fruit" + (f" {self.name!r}" if self.name != "" else "")
        ident += f" -> Features: {self.nfeatures()}"
        text = bar + "\n<" + f"{ident: ^78}" + ">\n" + bar + "\n"
        rule = "|" + 38 * "-"
        pairs = len(self._slices) - len(self._slices) % 2
        for i in range(0, pairs, 2):
            left = self._slices[i].summary().split("\n")
            right = self._slices[i + 1].summary().split("\n")
            height = max(len(left), len(right))
            left += [38 * " "] * (height - len(left))
            right += [38 * " "] * (height - len(right))
            text += rule + "|" + rule + "|\n"
            text += "\n".join(f"|{a}||{b}|" for a, b in zip(left, right))
            text += "\n" + rule + "|" + rule + "|\n"
        if len(self._slices) % 2:
            text += rule + "|\n|"
            text += self._slices[-1].summary().replace("\n", "|\n|")
            text += "|\n" + rule + "|\n"
        return text + bar

    def copy(self) -> "Fruit":
        dup = Fruit(self.name + " (Copy)")
        for slc in self._slices:
            dup.cut(slc.copy())
        return dup

    def deepcopy(self) -> "Fruit":
        dup = Fruit(self.name + " (Deepcopy)")
        for slc in self._slices:
            dup.cut(slc.deepcopy())
        return dup

    def label(self, index: int,
              level: Literal["prepared", "iterated sums", "features"] = "features",
              verbose: Literal[1, 2] = 1) -> str:
        """Label of one feature (or prepared input / iterated sum, by ``level``)."""
        for slc in self._slices:
            if level == "prepared":
                total = len(slc.get_preparateurs())
            elif level == "iterated sums":
                total = slc.niteratedsums()
            else:
                total = slc.nfeatures()
            if index < total:
                return slc.label(index, level, verbose)
            index -= total
        raise RuntimeError("Label index out of range")

    def __len__(self) -> int:
        return len(self._slices)

    def __iter__(self) -> "Fruit":
        self._cursor = -1
        return self

    def __next__(self) -> "FruitSlice":
        if self._cursor + 1 < len(self._slices):
            self._cursor += 1
            return self._slices[self._cursor]
        raise StopIteration()

    def __getitem__(self, index: int) -> "FruitSlice":
        return self.get_slice(index)


# The three stages of a slice, in pipeline order: (name of the stage's list, what may be added to
# it, the suffix of its add_ / get_ / clear_ verbs; fruits/fruit.py:280-429 is the contract - the
# verbs, what they accept and raise).  Everything a slice derives from its configuration (the
# fitted sieve copies, fused pipelines) is dropped whenever a stage changes.
_STAGES = (
    ("preparateurs", Preparateur, "preparateur", "preparateurs"),
    ("iss", ISS, "iss", "iss"),
    ("sieves", FeatureSieve, "sieve", "sieves"),
)


class FruitSlice:
    """One slice: preparateurs -> (chained) ISS -> sieves."""

    def __init__(self) -> None:
        self._stage: dict[str, list[Seed]] = {name: [] for name, *_ in _STAGES}
        self.fit_sample_size: Union[float, int] = 1
        self._invalidate()

    def _invalidate(self) -> None:
        # one list of fitted sieve copies per iterated sum
        self._sieves_extended: list[list[FeatureSieve]] = []
        self._fused_cache: dict = {}
        self._fitted: bool = False

    # the stage lists under the names the rest of the slice reads them by
    _preparateurs = property(lambda self: self._stage["preparateurs"])
    _iss = property(lambda self: self._stage["iss"])
    _sieves = property(lambda self: self._stage["sieves"])

    # ---- configuration ------------------------------------------------------
    def _put(self, stage: str, accepts: type, seed: Seed) -> None:
        if not isinstance(seed, accepts):
            raise TypeError
        self._stage[stage].append(seed)
        self._invalidate()

    def _wipe(self, stage: str) -> None:
        self._stage[stage] = []
        self._invalidate()

    def add(self, *objects: Union[Seed, Callable[[], Seed]]) -> None:
        """Appends seeds (instances, or classes to be instantiated without arguments) to the
        stage each belongs to."""
        for obj in objects:
            seed = obj() if inspect.isclass(obj) else obj
            for stage, accepts, _, _ in _STAGES:
                if isinstance(seed, accepts):
                    self._put(stage, accepts, seed)
                    break
            else:
                raise TypeError(f"Cannot add variable of type {type(seed)}")

    def clear(self) -> None:
        for stage, *_ in _STAGES:
            self._wipe(stage)
        self.fit_sample_size = 1

    def _fit_state(self) -> dict:
        return {"stage": self._stage, "extended": self._sieves_extended,
                "fit_sample_size": self.fit_sample_size}

    def _load_fit_state(self, state: dict) -> None:
        if [len(v) for v in state["stage"].values()] != [len(v) for v in self._stage.values()]:
            raise ValueError("the state belongs to a slice of another configuration")
        self._stage = state["stage"]
        self.fit_sample_size = state["fit_sample_size"]
        self._invalidate()
        self._sieves_extended = state["extended"]
        self._fitted = True

    def niteratedsums(self) -> int:
        return int(np.prod([iss.n_iterated_sums() for iss in self._iss]))

    def nfeatures(self) -> int:
        return self.niteratedsums() * sum(sieve.nfeatures() for sieve in self._sieves)

    def _compile(self) -> None:
        for stage, what in (("iss", "ISS"), ("sieves", "feature sieves")):
            if not self._stage[stage]:
                raise RuntimeError(f"No {what} given")

    def _select_fit_indices(self, X: np.ndarray) -> np.ndarray:
        # same draws from numpy's global generator as the reference
        # (fruits/fruit.py:430-438) so a seeded run picks the same series
        if isinstance(self.fit_sample_size, int) and self.fit_sample_size == 1:
            ind = np.random.randint(0, X.shape[0])
            return np.arange(ind, ind + 1)
        s = max(int(self.fit_sample_size * X.shape[0]), 1)
        return np.random.choice(X.shape[0], size=s, replace=False)

    def _select_fit_sample(self, X: np.ndarray) -> np.ndarray:
        return X[self._select_fit_indices(X), :, :]

    # ---- device pipeline --------------------------------------------------------
    def _prepare_device(self, Xd, cache, callbacks=(), fit_on=None):
        for prep in self._preparateurs:
            prep._cache = cache
            if fit_on is not None:
                prep.fit(nat.to_host(Xd) if prep._fit_needs_data() else fit_on)
            Xd = prep._transform_device(Xd)
            for cb in callbacks:
                cb.on_preparateur(nat.to_host(Xd))
        return Xd

    def _iterate_iss_device(self, Xd, iss_index: int = 0, upto: Optional[int] = None) -> Generator:
        """Yields every iterated sum as an (N, T) device tensor, in the order of
        fruits/fruit.py:440-454 (chained ISS feed each row to the next ISS).  ``upto``: stop in
        front of that ISS of the chain and yield its (N, 1, T) INPUTS instead."""
        if upto is not None and iss_index == upto:
            yield Xd
            return
        if iss_index == len(self._iss):
            yield Xd[:, 0, :]
            return
        iss = self._iss[iss_index]
        lookup = iss.lookup_device(Xd)
        for s, e in iss.word_batches(int(Xd.shape[0]), int(Xd.shape[2])):
            block = iss.transform_device(Xd, s, e, lookup)
            for k in range(block.shape[0]):
                if iss_index + 1 == len(self._iss):
                    yield block[k]
                else:
                    yield from self._iterate_iss_device(block[k].unsqueeze(1).contiguous(),
                                                        iss_index + 1, upto)

    # ---- device-side fit ------------------------------------------------------------
    def _fit_on_device(self, Sd, cache, deferred: Optional[list] = None) -> bool:
        """Fits the per-iterated-sum sieve copies without moving the fit sample's
        iterated sums to the host: the order statistics np.quantile interpolates
        between are selected on the device (fr_select_ranks).  Returns False when a
        sieve or the ISS layout needs the host path.  ``deferred``: the selections are only
        QUEUED (fr_select_ranks_begin); a closure per word batch that waits for its selection and
        forms the thresholds is appended - Fruit.fit runs them when every slice has been queued,
        so that the device goes from one slice's selection to the next slice's iterated sums
        without waiting for the host."""
        from .sieving.segment import SegmentSieve
        if os.environ.get("FRUITS_AMD_DEVICE_FIT", "1") == "0" or len(self._iss) != 1:
            return False
        for sv in self._sieves:
            if sv.requires_fitting and not (isinstance(sv, SegmentSieve)
                                            and 0 <= getattr(sv, "_inc", 0) <= 8):
                return False
        iss = self._iss[0]
        Ns, T = int(Sd.shape[0]), int(Sd.shape[2])
        n = Ns * T
        lookup = iss.lookup_device(Sd)
        self._sieves_extended = _FittedRows(self._sieves, cache)
        for s, e in iss.word_batches(Ns, T):
            block = iss.transform_device(Sd, s, e, lookup)
            # (what a sieve asks for depends on its q and the sample size only: once per sieve,
            # not once per copy)
            asks = [sv._quantile_requests(n) if sv.requires_fitting else None for sv in self._sieves]
            # Every iterated sum is asked for the same order statistics: the jobs of ONE row -
            # (differencing order, rank) pairs, those that several sieves share (NPI / MPI with the
            # same band) once - are laid out as a template and repeated for the rows with numpy
            # (19 000 jobs per fit of fruit_reduced: as Python loops their construction left the
            # device idle for 2-4 ms per slice between the iterated sums and their selection)
            t_index: dict = {}
            t_inc, t_rank, t_pairs = [], [], []
            for sv, reqs in zip(self._sieves, asks):
                if reqs is None:
                    continue
                idx = []
                for (_, lo, hi, _) in reqs:
                    for r in (lo, hi):
                        if (sv._inc, r) not in t_index:
                            t_index[(sv._inc, r)] = len(t_inc)
                            t_inc.append(sv._inc)
                            t_rank.append(r)
                    idx.append((t_index[(sv._inc, lo)], t_index[(sv._inc, hi)]))
                t_pairs.append((np.asarray([a for a, _ in idx], dtype=np.int64),
                                np.asarray([b for _, b in idx], dtype=np.int64)))
            K_rows, J = int(block.shape[0]), len(t_inc)
            rows = np.repeat(np.arange(K_rows, dtype=np.int32), J)
            incs = np.tile(np.asarray(t_inc, dtype=np.int32), K_rows)
            ranks = np.tile(np.asarray(t_rank, dtype=np.int64), K_rows)
            # the selection is QUEUED (fr_select_ranks_begin: no pass waits for the host) and this
            # thread goes on to the next slice; the thresholds of the rows are formed as arrays
            # when somebody waits for it (_FittedRows)
            pending = nat.Selection(block, rows, incs, ranks) if len(rows) else None
            tables = self._sieves_extended.add_rows(K_rows)

            def finish(pending=pending, block=block, t_pairs=t_pairs, tables=tables, asks=asks,
                       K_rows=K_rows, J=J):
                # (`block` lives until its selection is done)
                vals = (pending.result() if pending is not None else np.zeros(0)).reshape(K_rows, J)
                which = 0
                for i, (sieve, reqs) in enumerate(zip(self._sieves, asks)):
                    if reqs is not None:
                        lo_idx, hi_idx = t_pairs[which]
                        tables[i] = _interpolated_quantiles(sieve, reqs, vals[:, lo_idx], vals[:, hi_idx])
                        which += 1
            if deferred is None:
                finish()
            else:
                deferred.append(finish)
                # (a selection in flight owns one of 8 scratch blobs - and its (K, N, T) block:
                # a fruit of many slices, or a slice of many word batches, ends the oldest)
                while len(deferred) > _SELECTIONS_IN_FLIGHT:
                    deferred.pop(0)()
        return True

    # ---- fused ISS + sieves (one launch, no (K, N, T) tensor) --------------------
    def _fusable(self) -> bool:
        from .sieving.increment import MPI, NPI
        from .sieving.segment import END
        # (a chain of ISS: the LAST one fuses with the sieves, once per row of the chain in front
        # of it - fruits/fruit.py:440-454 feeds every row of an ISS to the next one)
        if os.environ.get("FRUITS_AMD_FUSED", "1") == "0" or not self._iss:
            return False
        last = self._iss[-1]
        if getattr(last, "_argmax", False):
            # Arctic argmax: the running maxima are materialised, every argmax row is formed in
            # LDS for the sieves that look at it (fr_pipeline_set_argmax) - differencing orders
            # 0 to 2, single ISS; anything else keeps fr_arctic_argmax + one launch per sieve
            if len(self._iss) != 1 or os.environ.get("FRUITS_AMD_FUSED_ARGMAX", "1") == "0":
                return False
            for sv in self._sieves:
                if type(sv) in (NPI, MPI) and not 0 <= sv._inc <= 2:
                    return False
        if type(last) is not ISS:
            from .iss.cos import CosWISS
            # the factorised CosWISS kernels fuse; the term-by-term path reduces first
            if type(last) is not CosWISS or not last._native():
                return False
            if last._ffn_size is not None and len(self._iss) != 1:
                return False     # (every (word, frequency) reads its own transformed input:
                                 # fused word by word, _transform_ffn_fused - single ISS only)
        for sv in self._sieves:
            if type(sv) not in (NPI, MPI, END):
                return False
            if type(sv) is not END and not -8 <= sv._inc <= 8:
                return False     # (cumulated rows: series of one time chunk; the pipeline says if not)
        return True

    def _fused(self, T: int, indices=None, chain_row: int = 0):
        """The fused pipeline for series length T (thresholds of the fitted sieve
        copies already resolved), or None when a sieve or the weighting is outside
        the fused set; cached until the next fit.  ``indices``: only these words
        (a rank's share of a word-sharded slice, fruits_amd.parallel); the pipeline
        then produces the feature columns of their iterated sums, in that order.
        ``chain_row`` (chained ISS): the pipeline of the LAST ISS for the ``chain_row``-th row of
        the chain in front of it - the same program, that row's thresholds."""
        key = T if indices is None else (T, tuple(indices))
        if chain_row:
            key = (key, chain_row)
        if key in self._fused_cache:
            return self._fused_cache[key]
        entry = None
        if self._fusable():
            from .sieving.segment import END
            iss = self._iss[-1]
            iss._check_supported()
            if hasattr(iss, "_arm_plan"):     # CosWISS dropout: the plan carries the mask
                iss._check_supported()
            argmax = None
            if getattr(iss, "_argmax", False):   # (the pipeline's rows are the argmax rows, not the plan's)
                argmax = [len(iss.words[i]) for i in (range(len(iss.words)) if indices is None else indices)]
            if indices is None:
                plan = iss._plan(0, len(iss.words))
                rows = range(plan.rows if argmax is None else iss._rows_of(0, len(iss.words)))
                if hasattr(iss, "_arm_plan"):
                    iss._arm_plan(plan, tuple(range(len(iss.words))), T)
            else:
                plan = iss._plan_indices(indices)
                if hasattr(iss, "_arm_plan"):
                    iss._arm_plan(plan, tuple(indices), T)
                first = np.concatenate([[0], np.cumsum(
                    [iss._depth(i) for i in range(len(iss.words))])]).astype(int)
                rows = [r for i in indices for r in range(first[i], first[i + 1])]
            if chain_row:
                rows = [chain_row * iss.n_iterated_sums() + r for r in rows]
            acting = (self._sieves_extended[rows[0]] if self._sieves_extended and len(rows)
                      else self._sieves)
            specs, cut_columns, n_slots = self._pipeline_specs(acting, T)
            try:
                pipe = (nat.Pipeline(plan, specs, T, argmax_lengths=argmax)
                        if len(rows) and plan.fits(T) else None)
            except ValueError:
                pipe = None
            if pipe is not None:
                assert pipe.rows == len(rows)
                quant = np.zeros((len(rows), pipe.q_stride))
                lazy = isinstance(self._sieves_extended, _FittedRows)
                if lazy:     # (the thresholds of all rows are arrays already)
                    off = 0
                    for i, sv in enumerate(self._sieves):
                        if type(sv) is END:
                            continue
                        th = self._sieves_extended.thresholds(i, rows)
                        if th is None:
                            sv._get_unfitted_quantiles()
                            th = sv._quantiles
                        quant[:, off:off + len(sv._q)] = th
                        off += len(sv._q)
                for k, row in enumerate(() if lazy else rows):
                    sieves = self._sieves_extended[row] if self._sieves_extended else self._sieves
                    off = 0
                    for sv in sieves:
                        if type(sv) is END:
                            continue
                        if not sv.requires_fitting:
                            sv._get_unfitted_quantiles()
                        quant[k, off:off + len(sv._q)] = sv._quantiles
                        off += len(sv._q)
                pipe.set_quantiles(quant)
                pipe._cut_columns = [(slot0, sv) for slot0, sv in cut_columns.values()]
                pipe._cut_slots = n_slots
                entry = pipe
        self._fused_cache[key] = entry
        return entry

    @staticmethod
    def _pipeline_specs(acting, T: int):
        """What fr_pipeline_create is told about the sieves ``acting`` for series of length T:
        (specs, cut columns, slots of the per-series cut table)."""
        from .sieving.segment import END
        # float ("coquantile") cuts differ from series to series: such a sieve names
        # columns of a per-series table instead of indices; sieves with the same cuts
        # share their columns (so NPI / MPI pairs still merge)
        # (the sieves that transform are the fitted COPIES, which forget coquantile_norm -
        # fruits/sieving/segment.py:90-91 - so the copies decide the norm here too)
        specs, cut_columns, n_slots = [], {}, 0
        for sv in acting:
            inc = 0 if type(sv) is END else sv._inc
            if sv._has_float_cuts():
                cuts_key = (tuple(sv._cut), sv._coquantile_norm)
                if cuts_key not in cut_columns:
                    cut_columns[cuts_key] = (n_slots, sv)
                    n_slots += len(sv._cut) + 1
                slot0 = cut_columns[cuts_key][0]
                specs.append((sv._kind | nat.FR_SIEVE_SERIES_CUTS, inc,
                              np.arange(slot0, slot0 + len(sv._cut) + 1), len(sv._q)))
            else:
                specs.append((sv._kind, inc, sv._int_cut_row(T), len(sv._q)))
        return specs, cut_columns, n_slots

    @staticmethod
    def _auto_prepare(pipe, N: int, T: int) -> None:
        """A fused launch asks for its pipeline's own kernels - the fused walk compiled at run time
        (hipRTC) with the sieves and the plan as immediates, 1.3 to 2 times as fast as the generic
        instance - WITHOUT ever waiting for the compiler: what an earlier process on this machine
        compiled comes from the disk cache right away (milliseconds; launches from 16 MiB of
        iterated sums on); a compilation (seconds the first time) runs on a helper thread for
        launches from 256 MiB on - this launch and any other before it is done take the generic
        kernel, later ones the compiled one (same results).  FRUITS_AMD_AUTO_PREPARE=0: never,
        =cached: what the disk cache and the kernels shipped with the build hold, no compiler,
        =all: every fused launch, and waited for; ``pipeline.prepare(N)`` is the explicit,
        synchronous way."""
        mode = os.environ.get("FRUITS_AMD_AUTO_PREPARE", "1")
        if mode == "0" or getattr(pipe, "_prepared_for", None) == N:
            return
        size = 8 * N * pipe.plan.rows * T
        if mode == "all":
            pipe.prepare(N)
        elif size >= (16 << 20):
            # (a stream capture must not see the uploads, module loads and the helper thread's
            # work: a captured launch takes what is there - fr_pipeline_prepare before the capture
            # is the documented way)
            if nat.torch().cuda.is_current_stream_capturing():
                return
            # the uploads (plan tables, the tables of a plan in pieces) and whatever the disk cache
            # or the kernels shipped with the build hold: here, on the caller's thread ...
            pipe.prepare_cached(N)
            # ... the helper thread only compiles and loads what is still missing
            if size >= (256 << 20) and mode != "cached" and not pipe.fully_compiled():
                pipe.prepare_in_background(N)
        else:
            return
        pipe._prepared_for = N

    def _arm_series_cuts(self, pipe, N: int, T: int, cache) -> None:
        """Uploads the per-series boundaries of the float-cut sieves of a fused pipeline
        (SegmentSieve._get_transformed_cuts on this batch's cache) for a run on N series."""
        if not getattr(pipe, "_cut_slots", 0):
            return
        table = np.zeros((N, pipe._cut_slots), dtype=np.int32)
        shape = np.empty((N, T))
        for first, sv in pipe._cut_columns:
            sv._cache = cache
            rows = sv._get_transformed_cuts(shape)      # sorted, leading 0; END checks its range
            table[:, first:first + rows.shape[1]] = np.clip(rows, 0, T)
        pipe.set_series_cuts(nat.to_device(table, dtype=np.int32))

    def _fusable_preparation(self, T: int):
        """(inc_lag, as_new, standardize, eps) when the preparateur chain is one the fused
        launch forms while staging the raw input - [INC | NEW(INC)] [STD], the chains of the
        experiment fruits - else None (the prepared input is then materialised)."""
        from .preparation.transform import INC, STD
        from .preparation.wrapper import NEW
        level = os.environ.get("FRUITS_AMD_FUSED_PREP", "1")
        if level == "0":
            return None
        # (CosWISS: the kernel CAN form the prepared rows where a letter reads them - but it reads
        # them once per (word, frequency, letter), so the differences and divisions are redone 165
        # times per series for fruit_reduced's slices: 3.08 ms against 2.53 ms with the small
        # prepared tensor materialised once.  Only on request: FRUITS_AMD_FUSED_PREP=2.)
        if level != "2" and any(type(iss) is not ISS for iss in self._iss):
            return None
        preps = list(self._preparateurs)
        lag, as_new, std, eps = 0, False, 0, 0.0

        def plain_inc(p):
            return (type(p) is INC and p._depth == 1 and p._zero_padding
                    and isinstance(p._lag(T), int) and 1 <= p._lag(T) < T)
        if preps and plain_inc(preps[0]):
            lag = preps.pop(0)._lag(T)
        elif preps and type(preps[0]) is NEW and preps[0]._preparateur is not None \
                and plain_inc(preps[0]._preparateur):
            lag, as_new = preps.pop(0)._preparateur._lag(T), True
        if preps and type(preps[0]) is STD and preps[0]._separately:
            p = preps.pop(0)
            std, eps = (2 if p._div_std else 1), float(p._eps)
        if preps or (lag == 0 and std == 0):
            return None
        # The lookup of a weighting is computed from what the ISS is handed
        # (fruits/iss/weighting.py:65-66) - the PREPARED input, which a fused preparation never
        # writes.  Safe are only weightings that do not look at the data (Indices, Plateaus) and
        # path lengths of the cache's RAW input; a Custom function, or any user subclass, gets
        # the materialised preparation.
        from .iss.weighting import _PathLength, _SeriesIndependent
        for iss in self._iss:
            w = getattr(iss, "weighting", None)
            if w is None or isinstance(w, _SeriesIndependent):
                continue
            if isinstance(w, _PathLength) and not w._on_prepared:
                continue
            return None
        return lag, as_new, std, eps

    def _attach(self, cache) -> None:
        for iss in self._iss:
            iss._cache = cache
            iss._attach_cache(None)

    # ---- fit / transform ----------------------------------------------------
    def fit(self, X: np.ndarray, cache: Optional[SharedSeedCache] = None,
            deferred: Optional[list] = None) -> None:
        self._compile()
        self._fused_cache = {}
        X = _check_batch(X)
        if cache is None:
            cache = SharedSeedCache(X)
        # the fit sample: rows of the input, gathered ON THE DEVICE from the cache's one upload of
        # X when the cache holds this batch (every slice of a fruit draws its own sample: a host
        # gather + upload per slice otherwise); the host copy only for a preparateur whose fit
        # looks at the data
        indices = self._select_fit_indices(X)
        needs_host = any(prep._fit_needs_data() for prep in self._preparateurs)
        whole = cache._input_dev is not None or 2 * len(indices) >= X.shape[0]   # (else: upload the sample alone)
        if cache._input is X and not needs_host and whole:
            sample = X[indices[:1], :, :]            # (what a data-blind Seed.fit is handed)
            Sd0 = cache.input_device(X)[nat.to_device(indices, dtype=np.int64)]
        else:
            sample = X[indices, :, :]
            Sd0 = nat.to_device(sample)
        Sd = self._prepare_device(Sd0, cache, fit_on=sample)
        self._attach(cache)
        for iss in self._iss:
            if iss.requires_fitting:
                iss.fit(nat.to_host(Sd))
        if not any(sieve.requires_fitting for sieve in self._sieves):
            self._fitted = True
            return
        self._sieves_extended = []
        if self._fit_on_device(Sd, cache, deferred):
            self._fitted = True
            return
        for itsum in self._iterate_iss_device(Sd):
            host = None
            fitted = [sieve.copy() for sieve in self._sieves]
            for sieve in fitted:
                sieve._cache = cache
                if sieve.requires_fitting:
                    if host is None:
                        host = nat.to_host(itsum)
                    sieve.fit(host)
            self._sieves_extended.append(fitted)
        self._fitted = True

    def transform(self, X: np.ndarray,
                  callbacks: Optional[list[AbstractCallback]] = None,
                  cache: Optional[SharedSeedCache] = None) -> np.ndarray:
        return nat.to_host(self.transform_device(X, callbacks, cache))

    def transform_device(self, X: np.ndarray,
                         callbacks: Optional[list[AbstractCallback]] = None,
                         cache: Optional[SharedSeedCache] = None):
        """``transform`` with the ``(N, nfeatures)`` result left on the device."""
        callbacks = callbacks or []
        if not self._fitted:
            raise RuntimeError("Missing call of self.fit")
        X = _check_batch(X)
        if cache is None:
            cache = SharedSeedCache(X)
        t = nat.torch()
        Xd = cache.input_device(X) if cache._input is X else nat.to_device(X)
        ffn = getattr(self._iss[-1], "_ffn_size", None) is not None
        if not callbacks and len(self._iss) == 1 and not ffn:
            # INC / NEW(INC) / STD formed while the fused launch stages the RAW rows: no
            # prepared tensor is written (one launch [+ the STD statistics pre-pass])
            T = int(Xd.shape[2])
            chain = self._fusable_preparation(T)
            fused = self._fused(T) if chain is not None else None
            if fused is not None:
                if getattr(fused, "_prep_chain", None) != (chain, int(Xd.shape[1])):
                    fused.set_preparation(int(Xd.shape[1]), chain[0], chain[1], chain[2], chain[3])
                    fused._prep_chain = (chain, int(Xd.shape[1]))
                if fused.raw_dims > 0:
                    self._attach(cache)
                    self._arm_series_cuts(fused, int(Xd.shape[0]), T, cache)
                    self._auto_prepare(fused, int(Xd.shape[0]), T)
                    return fused.run(Xd, self._iss[0].lookup_device(Xd))
        Pd = self._prepare_device(Xd, cache, callbacks)
        for cb in callbacks:
            cb.on_preparation_end(nat.to_host(Pd))
        self._attach(cache)
        if ffn and not callbacks and self._fusable():
            return self._transform_ffn_fused(Pd, cache)
        fused = None if callbacks else self._fused(int(Pd.shape[2]))
        if fused is not None and len(self._iss) > 1:
            return self._transform_chain_fused(Pd, cache)
        if fused is not None:
            if fused.raw_dims > 0:       # (was configured for raw input by another call)
                fused.set_preparation(int(Pd.shape[1]))
                fused._prep_chain = None
            self._arm_series_cuts(fused, int(Pd.shape[0]), int(Pd.shape[2]), cache)
            self._auto_prepare(fused, int(Pd.shape[0]), int(Pd.shape[2]))
            return fused.run(Pd, self._iss[0].lookup_device(Pd))
        feats = t.zeros((X.shape[0], self.nfeatures()), dtype=t.float64, device=Pd.device)
        col = 0
        for i, itsum in enumerate(self._iterate_iss_device(Pd)):
            for cb in callbacks:
                cb.on_iterated_sum(nat.to_host(itsum))
            sieves = self._sieves_extended[i] if self._sieves_extended else self._sieves
            for sieve in sieves:
                sieve._cache = cache
                nf = sieve.nfeatures()
                sieve.transform_device(itsum, feats, col)
                for cb in callbacks:
                    cb.on_sieve(nat.to_host(feats[col:col + nf]))
                col += nf
        if callbacks:
            out = nat.to_host(feats)
            for cb in callbacks:
                cb.on_sieving_end(out)
        return feats

    def _transform_chain_fused(self, Pd, cache):
        """Chained ISS (fruits/fruit.py:440-454): the rows of the chain in front of the last ISS
        are materialised one word batch at a time, and every one of them goes through ONE fused
        launch of the last ISS + the sieves (that row's thresholds) - K_chain launches instead of
        K_chain x K_last x |sieves|, and no (K_chain x K_last, N, T) tensor."""
        t = nat.torch()
        N, T = int(Pd.shape[0]), int(Pd.shape[2])
        last = self._iss[-1]
        width = last.n_iterated_sums() * sum(sv.nfeatures() for sv in self._sieves)
        feats = t.empty((N, self.nfeatures()), dtype=t.float64, device=Pd.device)
        for r, Xin in enumerate(self._iterate_iss_device(Pd, 0, upto=len(self._iss) - 1)):
            pipe = self._fused(T, chain_row=r)
            if pipe.raw_dims > 0:
                pipe.set_preparation(1)
            self._arm_series_cuts(pipe, N, T, cache)
            feats[:, r * width:(r + 1) * width] = pipe.run(Xin, last.lookup_device(Xin))
        return feats

    def _transform_ffn_fused(self, Pd, cache, words=None):
        """CosWISS with the randomised ffn (fruits/iss/cos.py:93-137): every (word, frequency)
        reads its own transformed copy of the input, so the slice runs word by word - the F copies
        of a word (coswiss_ffn) go through ONE fused launch of that word's units and the sieves
        (pipeline over the word alone, its rows' thresholds): W launches, and neither the
        (W x F, N, T) iterated sums nor a sieve pass over them.  ``words``: only these (a rank's
        share of a word-sharded slice), their column blocks in that order."""
        t = nat.torch()
        iss = self._iss[0]
        N, D, T = (int(v) for v in Pd.shape)
        F = len(iss._freqs)
        width = F * sum(sv.nfeatures() for sv in self._sieves)
        words = list(range(len(iss.words))) if words is None else list(words)
        feats = t.empty((N, len(words) * width), dtype=t.float64, device=Pd.device)
        Z = t.empty((F, N, D, T), dtype=t.float64, device=Pd.device)
        for slot, w in enumerate(words):
            pipe = self._fused(T, indices=(w,))
            if pipe.plan.max_dim > D:
                raise IndexError(f"a word references dimension {pipe.plan.max_dim} but "
                                 f"the input has only {D}")
            for f in range(F):
                nat.coswiss_ffn(Pd, iss._A[w, f], iss._b[w, f], iss._C[w, f], Z[f])
            nat.check(nat.lib().fr_coswiss_set_input_stride(pipe.plan._h, N * D * T))
            self._arm_series_cuts(pipe, N, T, cache)
            feats[:, slot * width:(slot + 1) * width] = pipe.run(Z[0], None)
        return feats

    def fit_transform(self, X: np.ndarray) -> np.ndarray:
        self.fit(X)
        return self.transform(X)

    # ---- introspection ------------------------------------------------------
    def summary(self) -> str:
        def line(s: str) -> str:
            return f"{s: <38}"

        rows = [f"{f'FruitSlice -> {self.nfeatures()}': ^38}", 38 * "-",
                line(f"Preparateurs ({len(self._preparateurs)}):")]
        text = "\n".join(rows) + "\n"
        text += "\n".join(line(f"    + {p}") for p in self._preparateurs)
        if not self._preparateurs:
            text += 38 * " "
        text += "\n" + line(f"ISS Calculators ({len(self._iss)}):")
        if not self._iss:
            text += 38 * " "
        for iss in self._iss:
            text += "\n" + line(f"    + {iss} -> {iss.n_iterated_sums()}")
            text += "\n" + line(f"       | words: {len(iss.words)}")
            text += "\n" + line(f"       | semiring: {iss.semiring.__class__.__name__}")
            wname = "None" if iss.weighting is None else iss.weighting.__class__.__name__
            text += "\n" + line(f"       | weighting: {wname}")
        if not self._iss:
            text += "\n"
        text += "\n" + line(f"Sieves ({len(self._sieves)}):")
        if not self._sieves:
            text += "\n" + 38 * " "
        for sv in self._sieves:
            text += "\n" + line(f"    + {sv.__class__.__name__} -> {sv.nfeatures()}")
        return text

    def copy(self) -> "FruitSlice":
        dup = FruitSlice()
        dup.add(*self._preparateurs, *self._iss, *self._sieves)
        return dup

    def deepcopy(self) -> "FruitSlice":
        dup = FruitSlice()
        for seed in (*self._preparateurs, *self._iss, *self._sieves):
            dup.add(seed.copy())
        dup.fit_sample_size = self.fit_sample_size
        return dup

    def label(self, index: int,
              level: Literal["prepared", "iterated sums", "features"] = "features",
              verbose: Literal[1, 2] = 1) -> str:
        def short(label: str, sep: str) -> str:
            return label.split(sep)[0] if verbose == 1 else label

        if level == "prepared":
            parts = [short(p.label(), "(") for p in self._preparateurs[:index + 1]]
            return " -> ".join(parts) if parts else "input"
        text = " -> ".join(short(p.label(), "(") for p in self._preparateurs)
        if text:
            text += " | "
        findex = 0
        if level == "features":
            per_sum = int(np.sum([s.nfeatures() for s in self._sieves]))
            index, findex = map(int, divmod(index, per_sum))
        remaining = self.niteratedsums()
        words = []
        for iss in self._iss:
            k = iss.n_iterated_sums()
            words.append(short(iss.label(int((index % remaining) // (remaining / k))), " : "))
            remaining /= k
        text += " -> ".join(words)
        if level == "iterated sums":
            return text if text else "input"
        if text:
            text += " | "
        for sieve in self._sieves:
            if findex < sieve.nfeatures():
                return text + sieve.label(findex)
            findex -= sieve.nfeatures()
        raise RuntimeError("Feature index out of range")


def _stage_verbs(stage: str, accepts: type, one: str, many: str) -> None:
    """add_<one> / get_<many> / clear_<many> of one stage of a slice."""
    def add(self, seed) -> None:
        self._put(stage, accepts, seed)

    def get(self) -> list:
        return self._stage[stage]

    def clear(self) -> None:
        self._wipe(stage)

    for verb, fn in ((f"add_{one}", add), (f"get_{many}", get), (f"clear_{many}", clear)):
        fn.__name__ = fn.__qualname__ = verb
        setattr(FruitSlice, verb, fn)


for _row in _STAGES:
    _stage_verbs(*_row)

