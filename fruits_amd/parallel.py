"""Multi-GPU execution: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The ISS path shards on two independent axes and needs no exchange inside the
scan:

* **series** (``shard_series``): every rank transforms its own rows and returns
  its own feature rows - no collective at all.  This is what ``bench.py --gpus N``
  measures (weak scaling).
* **words** (``transform_sharded``): every rank holds the whole batch and owns a
  balanced share of the word list's first-letter sub-tries, i.e. of the iterated
  sums and therefore of the feature COLUMNS; one all-gather of the ``(N, F_r)``
  feature blocks (padded to the widest block - RCCL has no all-gatherv) plus a
  column permutation restores the reference's column order.  On a fully
  connected 8-GPU xGMI node each block crosses each link once.

The shard planning and the gather / permutation are plain host logic and are
tested on CPU with the ``gloo`` backend; the compute callback is the HIP
pipeline on GPU ranks.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np


# ----------------------------------------------------------------------- series axis
def shard_series(n_series: int, rank: int, world: int) -> slice:
    """Contiguous, balanced block of series for ``rank`` (first ranks get the
    remainder)."""
    base, rem = divmod(n_series, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


# ----------------------------------------------------------------------- word axis
def _letters(word_string: str) -> list[str]:
    return [p + "]" for p in word_string.split("]")[:-1]]


def shard_words(word_strings: Sequence[str], depths: Sequence[int], world: int):
    """Assigns sub-tries of the word list to ranks.

    Words that share a first letter start out together (their common prefixes
    are then computed once, on one rank).  While the heaviest sub-trie is more
    than its fair share it is split one letter deeper - the ranks that receive
    the pieces recompute the short common prefix, which is cheap next to the
    imbalance.  Sub-tries are placed heaviest-first on the lightest rank; weight
    = scan passes + output rows.  Returns, per rank, the sorted word indices."""
    letters = [_letters(s) for s in word_strings]
    cost = [len(letters[i]) + depths[i] for i in range(len(word_strings))]
    groups: dict[tuple, list[int]] = {}
    for i, ls in enumerate(letters):
        groups.setdefault(tuple(ls[:1]), []).append(i)
    total = sum(cost)
    fair = 1.25 * total / max(world, 1)
    while True:
        heavy = max(groups, key=lambda k: (sum(cost[i] for i in groups[k]), k), default=None)
        if heavy is None or sum(cost[i] for i in groups[heavy]) <= fair:
            break
        depth = len(heavy) + 1
        pieces: dict[tuple, list[int]] = {}
        for i in groups[heavy]:
            pieces.setdefault(tuple(letters[i][:depth]), []).append(i)
        if len(pieces) <= 1:
            break                      # a single chain cannot be split further
        del groups[heavy]
        for k, v in pieces.items():
            groups.setdefault(k, []).extend(v)
    weight = {k: sum(cost[i] for i in v) for k, v in groups.items()}
    order = sorted(groups, key=lambda k: (-weight[k], groups[k][0]))
    load = [0] * world
    parts: list[list[int]] = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += weight[k]
        parts[r].extend(groups[k])
    return [sorted(p) for p in parts]


def row_ranges(depths: Sequence[int]):
    """Global iterated-sum row range [begin, end) of every word (reference
    order: words in order, shortest emitted prefix first)."""
    out, k = [], 0
    for d in depths:
        out.append((k, k + d))
        k += d
    return out


def column_map(parts, depths, features_per_sum: int):
    """For every rank the global feature columns its local block maps to, in
    local column order."""
    rr = row_ranges(depths)
    maps = []
    for words in parts:
        cols = []
        for i in words:
            for row in range(*rr[i]):
                cols.extend(range(row * features_per_sum, (row + 1) * features_per_sum))
        maps.append(np.asarray(cols, dtype=np.int64))
    return maps


def gather_index(maps, n_features: int):
    """The concatenated column map of a word-sharded gather: for every feature column the rank
    that computed it and its position in that rank's block."""
    rank_of = np.full(n_features, -1, dtype=np.int64)
    pos_of = np.zeros(n_features, dtype=np.int64)
    for r, m in enumerate(maps):
        rank_of[m] = r
        pos_of[m] = np.arange(len(m))
    if (rank_of < 0).any():
        raise ValueError("the column maps do not cover every feature column")
    return rank_of, pos_of


def gather_features(local, maps, n_features: int, rank: int, world: int, group=None,
                    timings: Optional[dict] = None):
    """All-gathers the per-rank ``(N, F_r)`` blocks and moves their columns to the reference
    positions - ONE indexed gather over the concatenated column map, whatever the world size.
    ``local`` is a torch tensor: a device tensor with the nccl (= RCCL) backend; with gloo (CPU
    rehearsals of the same path) a device tensor is staged through the host for the collective
    only.  ``timings``: filled with the seconds spent in the collective (synchronised)."""
    import time

    import torch
    import torch.distributed as dist

    width = max(len(m) for m in maps)
    n = local.shape[0]
    padded = torch.zeros((n, width), dtype=local.dtype, device=local.device)
    padded[:, :local.shape[1]] = local
    if world > 1:
        via_host = local.is_cuda and dist.get_backend(group) == "gloo"
        if timings is not None and local.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if via_host:
            flat_h = torch.empty((world * n, width), dtype=local.dtype)
            dist.all_gather_into_tensor(flat_h, padded.cpu(), group=group)
            flat = flat_h.to(local.device)
        else:
            flat = torch.empty((world * n, width), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(flat, padded, group=group)
        if timings is not None:
            if local.is_cuda:
                torch.cuda.synchronize()
            timings["allgather_s"] = time.perf_counter() - t0
            timings["allgather_bytes_per_rank"] = int(n * width * local.element_size())
        gathered = flat.view(world, n, width)
    else:
        gathered = padded.unsqueeze(0)
    rank_of, pos_of = gather_index(maps, n_features)
    r_idx = torch.as_tensor(rank_of, device=local.device)
    p_idx = torch.as_tensor(pos_of, device=local.device)
    # gathered[rank_of[c], :, pos_of[c]] for every column c: (F, N), then the transpose
    return gathered[r_idx, :, p_idx].t().contiguous()


# ----------------------------------------------------------------------- the batch, from the root
def _rank_world(rank, world, group):
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    return rank, world


def broadcast_batch(X, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                    root: int = 0, device: Optional[bool] = None):
    """The batch every rank of a word-sharded transform needs, from ``root`` alone (the other
    ranks pass ``None``): uploaded ONCE, on the root, and broadcast device to device (RCCL over
    xGMI with the nccl backend; a gloo group stages through the host).  ``X``: a float64
    ``(N, D, T)`` numpy array or a device tensor.  ``device``: whether the result lives on the
    GPU (default: when one is there).  Returns a torch tensor."""
    import torch
    import torch.distributed as dist

    rank, world = _rank_world(rank, world, group)
    if device is None:
        device = torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if device else torch.device("cpu")
    if rank == root:
        if X is None:
            raise ValueError("the root rank has to hand in the batch")
        Xb = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X))
        if Xb.dtype != torch.float64 or Xb.dim() != 3:
            raise TypeError("input has to be a float64 array of shape (N, D, T)")
        Xb = Xb.to(dev).contiguous()
    if world == 1:
        return Xb
    shape = [tuple(Xb.shape) if rank == root else None]
    dist.broadcast_object_list(shape, src=root, group=group)
    if rank != root:
        Xb = torch.empty(shape[0], dtype=torch.float64, device=dev)
    if Xb.is_cuda and dist.get_backend(group) != "nccl":
        host = Xb.cpu()
        dist.broadcast(host, src=root, group=group)
        if rank != root:
            Xb.copy_(host)
    else:
        dist.broadcast(Xb, src=root, group=group)
    return Xb


# ----------------------------------------------------------------------- fruit slice
def slice_transform_sharded(slc, Xb, cache, rank: int, world: int, group=None,
                            block: Optional[Callable] = None, timings: Optional[dict] = None):
    """``FruitSlice.transform`` with the slice's word list sharded over ``world`` ranks; ``Xb``
    is the batch as a torch tensor (``broadcast_batch``).  ``block(slc, iss, Xb, cache, word
    indices, depths, features per sum) -> (N, F_r)`` produces the local block; the default runs
    the HIP pipeline.  Returns the ``(N, F)`` features as a torch tensor on ``Xb``'s device.
    Slices with chained ISS fall back to replicated execution."""
    t = Xb.new_empty(0)
    if len(slc._iss) != 1:
        X = Xb.cpu().numpy()
        return slc.transform_device(X, cache=None).to(t.device)
    iss = slc._iss[0]
    strings = [str(w) for w in iss.words]
    depths = [iss._depth(i) for i in range(len(strings))]
    per_sum = sum(s.nfeatures() for s in slc._sieves)
    parts = shard_words(strings, depths, world)
    maps = column_map(parts, depths, per_sum)
    if timings is None:
        local = (block or _device_block)(slc, iss, Xb, cache, parts[rank], depths, per_sum)
        return gather_features(local, maps, slc.nfeatures(), rank, world, group)
    import time
    import torch
    sync = torch.cuda.synchronize if Xb.is_cuda else (lambda: None)
    sync()
    t0 = time.perf_counter()
    local = (block or _device_block)(slc, iss, Xb, cache, parts[rank], depths, per_sum)
    sync()
    t1 = time.perf_counter()
    full = gather_features(local, maps, slc.nfeatures(), rank, world, group)
    sync()
    timings["compute_s"] = timings.get("compute_s", 0.0) + t1 - t0
    timings["gather_s"] = timings.get("gather_s", 0.0) + time.perf_counter() - t1
    timings["gathered_bytes_per_rank"] = (timings.get("gathered_bytes_per_rank", 0)
                                          + int(Xb.shape[0]) * max(len(m) for m in maps) * 8)
    return full


def _device_block(slc, iss, Xd, cache, indices, depths, per_sum):
    """The feature columns of the words ``indices`` of a slice on the device batch ``Xd``."""
    from . import _native as nat
    t = nat.torch()
    N = int(Xd.shape[0])
    rr = row_ranges(depths)
    n_rows = sum(depths[i] for i in indices)
    if not indices:
        return t.zeros((N, 0), dtype=t.float64, device=Xd.device)
    # the rank's share in ONE launch on the RAW batch where the slice's preparation fuses into
    # the staging (INC / NEW(INC) / STD): as in FruitSlice.transform_device
    T = int(Xd.shape[2])
    # (a CosWISS with the randomised ffn reads a transformed copy of the input per (word,
    # frequency) - fruits/iss/cos.py:93-137: its pipelines must not be run on the plain batch;
    # the rank's words go through FruitSlice._transform_ffn_fused's word-by-word launches)
    ffn = getattr(iss, "_ffn_size", None) is not None
    chain = None if ffn else slc._fusable_preparation(T)
    fused = slc._fused(T, indices=indices) if chain is not None else None
    if fused is not None and fused.set_preparation(int(Xd.shape[1]), *chain):
        slc._attach(cache)
        slc._arm_series_cuts(fused, N, T, cache)
        slc._auto_prepare(fused, N, T)
        return fused.run(Xd, iss.lookup_device(Xd))
    Pd = slc._prepare_device(Xd, cache)
    slc._attach(cache)
    if ffn and slc._fusable():
        return slc._transform_ffn_fused(Pd, cache, words=indices)
    fused = None if ffn else slc._fused(int(Pd.shape[2]), indices=indices)
    if fused is not None:      # the rank's share in ONE launch, no (K_r, N, T) tensor
        fused.set_preparation(int(Pd.shape[1]))     # (prepared input: nothing to fuse)
        slc._arm_series_cuts(fused, int(Pd.shape[0]), int(Pd.shape[2]), cache)
        slc._auto_prepare(fused, int(Pd.shape[0]), int(Pd.shape[2]))
        return fused.run(Pd, iss.lookup_device(Pd))
    feats = t.zeros((N, n_rows * per_sum), dtype=t.float64, device=Pd.device)
    block = iss.transform_device(Pd, indices=indices)
    col = k = 0
    for i in indices:
        for row in range(*rr[i]):
            sieves = slc._sieves_extended[row] if slc._sieves_extended else slc._sieves
            for sieve in sieves:
                sieve._cache = cache
                sieve.transform_device(block[k], feats, col)
                col += sieve.nfeatures()
            k += 1
    return feats


def transform_sharded(fruit, X=None, rank: Optional[int] = None, world: Optional[int] = None,
                      group=None, root: int = 0, on_device: bool = False,
                      block: Optional[Callable] = None, timings: Optional[dict] = None):
    """``Fruit.transform`` with every slice's words sharded over the ranks of the (default)
    process group - north_star's split: every rank holds the batch, computes the feature columns
    of its sub-tries in one fused launch per slice, one all-gather per slice (RCCL over xGMI)
    and a column gather assemble the reference's ``(N, nfeatures)`` on every rank.

    ``X`` is looked at on ``root`` only (the other ranks may pass ``None``): it is uploaded once
    there and broadcast device to device.  A fruit that is not fitted yet is fitted on the
    root (``fit_on_root``: the others receive the fitted state, not the data's fit).  The
    features stay on the device until the end; ``on_device``: return the device tensor instead
    of a numpy array.  ``timings``: filled with the seconds of the broadcast, of this rank's
    launches and of the all-gathers + column gathers (each synchronised: for benchmarks)."""
    import time

    import torch
    from .cache import SharedSeedCache

    def tick():
        if timings is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.perf_counter()

    rank, world = _rank_world(rank, world, group)
    if world > 1:
        import torch.distributed as dist
        flags = [bool(fruit._fitted)]
        dist.broadcast_object_list(flags, src=root, group=group)
        root_fitted = flags[0]
    else:
        root_fitted = bool(fruit._fitted)
    if not root_fitted:
        if rank == root and not isinstance(X, np.ndarray):
            raise TypeError("fitting needs the batch as a numpy array on the root")
        fit_on_root(fruit, X if rank == root else None, rank, world, group, root)
    elif not fruit._fitted:
        raise RuntimeError("Missing call of self.fit (the root's fruit is fitted, this rank's is not: "
                           "fit with fruits_amd.parallel.fit_on_root)")
    t0 = tick()
    Xb = broadcast_batch(X, rank, world, group, root)
    t1 = tick()
    cache = SharedSeedCache(X if isinstance(X, np.ndarray) and rank == root else None)
    if Xb.is_cuda:
        cache.adopt_device_input(Xb)
    if timings is not None:
        timings.update(broadcast_s=t1 - t0, compute_s=0.0, gather_s=0.0)
    blocks = [slice_transform_sharded(slc, Xb, cache, rank, world, group, block, timings)
              for slc in fruit._slices]
    result = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)
    result = torch.nan_to_num(result, nan=0.0, posinf=None, neginf=None) if not result.is_cuda \
        else _nan_to_num_device(result)
    return result if on_device else result.cpu().numpy()


def _nan_to_num_device(t):
    from . import _native as nat
    return nat.nan_to_num(t.contiguous())


# ----------------------------------------------------------------------- fit once, series axis
def fit_on_root(fruit, X, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                root: int = 0) -> None:
    """``Fruit.fit`` on ONE rank, the fitted state (thresholds of every sieve copy, statistics of
    the preparateurs, random state of a CosWISS) broadcast to the others - instead of every rank
    repeating the same fit on the same data.  ``X`` is only looked at on ``root``."""
    import torch.distributed as dist

    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    box = [None]
    if rank == root:
        fruit.fit(X)
        box[0] = fruit.fit_state()
    if world > 1:
        dist.broadcast_object_list(box, src=root, group=group)
    if rank != root:
        fruit.load_fit_state(box[0])


def transform_series_sharded(fruit, X: np.ndarray, rank: Optional[int] = None,
                             world: Optional[int] = None, group=None, gather: bool = True):
    """``Fruit.transform`` with the SERIES sharded over the ranks: every rank transforms its own
    contiguous block of rows with the whole word list - no collective on the data path.
    ``gather``: the blocks are then all-gathered (padded to the largest block) so that every
    rank returns the full ``(N, nfeatures)``; else a rank returns its own rows."""
    import torch
    import torch.distributed as dist

    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = shard_series(X.shape[0], rank, world)
    local = fruit.transform(np.ascontiguousarray(X[mine]))
    if not gather or world == 1:
        return local
    rows = [shard_series(X.shape[0], r, world) for r in range(world)]
    tallest = max(sl.stop - sl.start for sl in rows)
    padded = torch.zeros((tallest, local.shape[1]), dtype=torch.float64)
    padded[:local.shape[0]] = torch.from_numpy(local)
    flat = torch.empty((world * tallest, local.shape[1]), dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        flat_d = flat.to(dev)
        dist.all_gather_into_tensor(flat_d, padded.to(dev), group=group)
        flat = flat_d.cpu()
    else:
        dist.all_gather_into_tensor(flat, padded, group=group)
    blocks = flat.view(world, tallest, local.shape[1]).numpy()
    return np.concatenate([blocks[r, :rows[r].stop - rows[r].start] for r in range(world)], axis=0)

