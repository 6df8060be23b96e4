"""Multi-rank host logic on CPU: word sharding, all-gather (gloo, world_size 2)
and the column permutation, with the oracle standing in for the local compute;
plus the series partition."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from fruits_amd import parallel as par
from oracle import ref_numpy as orc

G = load_golden()


def test_shard_series_partition():
    for n, w in [(2048, 8), (10, 3), (5, 8), (0, 2)]:
        parts = [par.shard_series(n, r, w) for r in range(w)]
        idx = np.concatenate([np.arange(n)[p] for p in parts])
        np.testing.assert_array_equal(idx, np.arange(n))
        sizes = [p.stop - p.start for p in parts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("key,world", [("2,3", 2), ("4,2", 4), ("6,2", 8), ("9,1", 8), ("2,3", 8)])
def test_shard_words_is_a_balanced_partition(key, world):
    ent = G.manifest["words"][key]
    parts = par.shard_words(ent["words"], ent["plan"], world)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(len(ent["words"])))
    maps = par.column_map(parts, ent["plan"], 2)
    cols = np.sort(np.concatenate(maps))
    np.testing.assert_array_equal(cols, np.arange(2 * ent["K"]))
    if len(ent["words"]) >= 4 * world:
        loads = [sum(ent["plan"][i] + ent["words"][i].count("[") for i in p) for p in parts]
        assert max(loads) <= 1.6 * (sum(loads) / world), loads


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_block(X, words, depths, indices, q_med):
    """Local feature block [NPI(q=(q,inf)) count, END] per iterated sum of the shard."""
    plan_rows = par.row_ranges(depths)
    its = orc.iss_transform(X, words, "EXTENDED")
    cols = []
    for i in indices:
        for row in range(*plan_rows[i]):
            inc = orc.pre_transform(its[row], 1)
            cols.append(np.sum(inc > q_med, axis=1))
            cols.append(its[row][:, -1])
    if not cols:
        return torch.zeros((X.shape[0], 0), dtype=torch.float64)
    return torch.from_numpy(np.stack(cols, axis=1).astype(np.float64))


def _worker(rank, world, port, key, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ent = G.manifest["words"][key]
        X = np.random.default_rng(7).random((6, 3, 48))
        parts = par.shard_words(ent["words"], ent["plan"], world)
        maps = par.column_map(parts, ent["plan"], 2)
        local = _oracle_block(X, ent["words"], ent["plan"], parts[rank], 0.25)
        full = par.gather_features(local, maps, 2 * ent["K"], rank, world)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("key", ["2,3", "3,3"])
def test_word_sharded_gather_matches_unsharded(tmp_path, key):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, key, str(tmp_path)), nprocs=world, join=True)
    ent = G.manifest["words"][key]
    X = np.random.default_rng(7).random((6, 3, 48))
    ref = _oracle_block(X, ent["words"], ent["plan"], list(range(len(ent["words"]))), 0.25).numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        np.testing.assert_array_equal(got, ref)   # bit-for-bit: same arithmetic, only moved


def test_single_rank_gather_is_identity():
    ent = G.manifest["words"]["2,3"]
    parts = par.shard_words(ent["words"], ent["plan"], 1)
    maps = par.column_map(parts, ent["plan"], 3)
    local = torch.arange(4 * 54, dtype=torch.float64).reshape(4, 54)
    out = par.gather_features(local, maps, 54, 0, 1)
    torch.testing.assert_close(out, local)


class _StandInFruit:
    """Host stand-in with the interface the multi-rank helpers use (the real Fruit computes on
    the GPU): `fit` takes a threshold from the data, `transform` is the oracle's block."""

    def __init__(self, key):
        self.ent = G.manifest["words"][key]
        self.q = None
        self.fits = 0

    def fit(self, X):
        self.fits += 1
        self.q = float(np.median(X))

    def fit_state(self):
        import pickle
        return pickle.dumps(self.q)

    def load_fit_state(self, state):
        import pickle
        self.q = pickle.loads(state)

    def transform(self, X):
        idx = list(range(len(self.ent["words"])))
        return _oracle_block(X, self.ent["words"], self.ent["plan"], idx, self.q).numpy()


def _series_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = np.random.default_rng(11).random((7, 3, 40))
        fruit = _StandInFruit("2,3")
        # only the root looks at the fit data
        par.fit_on_root(fruit, X if rank == 0 else None)
        assert fruit.fits == (1 if rank == 0 else 0)
        full = par.transform_series_sharded(fruit, X)
        own = par.transform_series_sharded(fruit, X, gather=False)
        np.save(os.path.join(out_dir, f"full{rank}.npy"), full)
        np.save(os.path.join(out_dir, f"own{rank}.npy"), own)
    finally:
        dist.destroy_process_group()


def test_fit_on_root_and_series_sharded_transform(tmp_path):
    """fit once, broadcast the fitted state; every rank transforms its rows (uneven blocks) and
    the gathered matrix equals the unsharded transform on every rank."""
    world = 2
    mp.spawn(_series_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    X = np.random.default_rng(11).random((7, 3, 40))
    fruit = _StandInFruit("2,3")
    fruit.fit(X)
    ref = fruit.transform(X)
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"full{r}.npy")), ref)
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"own{r}.npy")),
                                      ref[par.shard_series(7, r, world)])


def test_fit_state_round_trip():
    """The fitted state of a fruit is picklable (no device handles, no cache) and makes an
    equally configured fruit the fitted one."""
    import fruits_amd as fr

    def build():
        fruit = fr.Fruit("state")
        fruit.add(fr.preparation.STD(separately=False), fr.ISS(fr.words.of_weight(2, dim=2)))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
        return fruit
    a, b = build(), build()
    slc = a.get_slice()
    slc.get_preparateurs()[0]._mean, slc.get_preparateurs()[0]._std = 0.25, 1.5
    slc.get_iss()[0]._plans["x"] = object()          # a device handle must not travel
    rows = []
    for k in range(slc.niteratedsums()):
        copies = [sv.copy() for sv in slc.get_sieves()]
        copies[0]._quantiles = np.array([0.1 * k, np.inf])
        rows.append(copies)
    slc._sieves_extended = rows
    slc._fitted = a._fitted = True
    with pytest.raises(RuntimeError):
        b.fit_state()
    b.load_fit_state(a.fit_state())
    got = b.get_slice()
    assert b._fitted and got._fitted and got.get_iss()[0]._plans == {}
    assert got.get_preparateurs()[0]._std == 1.5
    assert len(got._sieves_extended) == slc.niteratedsums()
    np.testing.assert_array_equal(got._sieves_extended[3][0]._quantiles, [0.1 * 3, np.inf])
    with pytest.raises(ValueError):
        fr.Fruit("other").load_fit_state(a.fit_state())


@pytest.mark.parametrize("variant", ["terms", "ffn", "dropout"])
def test_fit_state_round_trip_coswiss(variant):
    """A CosWISS keeps device programs of two kinds (factorised plans, and the term programs of
    the non-factorised path: ctypes handles + device tensors); neither is part of the pickled
    state, and the receiving side starts with empty tables of both."""
    import pickle
    import fruits_amd as fr
    from fruits_amd import _native as nat
    words = [fr.words.SimpleWord(s) for s in ["[1]", "[1][2]"]]
    kw = {"terms": dict(exponent=nat.CosPlan.MAX_EXPONENT + 1), "ffn": dict(ffn_size=3),
          "dropout": dict(dropout=0.5)}[variant]
    cos = fr.CosWISS(words, freqs=[0.2, 0.7], **kw)
    cos._programs[(0, 2, 2)] = (C_handle := object(), "device tensors")
    cos._plans[("cos", (0, 1))] = C_handle
    if variant == "ffn":
        np.random.seed(0)
        cos._A = np.random.rand(2, 2, 3, 2)           # (the fitted weights DO travel)
    back = pickle.loads(pickle.dumps(cos))
    assert back._programs == {} and back._plans == {}
    assert back._exponent == cos._exponent and list(back._freqs) == [0.2, 0.7]
    if variant == "ffn":
        np.testing.assert_array_equal(back._A, cos._A)
    fruit = fr.Fruit("cos")
    fruit.add(cos, fr.sieving.END)
    fruit._fitted = fruit.get_slice()._fitted = True
    pickle.loads(pickle.dumps(fruit.fit_state()))


def _sharded_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fruits_amd as fr
        words = fr.words.of_weight(3, dim=3)
        fruit = fr.Fruit("sharded")
        fruit.add(fr.preparation.INC, fr.ISS(words, mode=fr.ISSMode.EXTENDED), fr.sieving.NPI, fr.sieving.END)
        fruit.cut()
        fruit.add(fr.ISS(fr.words.of_weight(2, dim=3), mode=fr.ISSMode.EXTENDED), fr.sieving.NPI, fr.sieving.END)
        fruit._fitted = True                       # (nothing to fit: NPI() counts positive increments)
        for slc in fruit:
            slc._fitted = True
        X = np.random.default_rng(13).random((5, 3, 40)) if rank == 0 else None
        seen = {}

        def block(slc, iss, Xb, cache, indices, depths, per_sum):
            # the oracle stands in for the HIP pipeline; what is under test is what surrounds it:
            # only the root handed the batch in, every rank sees it here
            seen["shape"] = tuple(Xb.shape)
            strings = [str(w) for w in iss.words]
            return _oracle_block(Xb.numpy(), strings, depths, indices, 0.0)
        full = par.transform_sharded(fruit, X, block=block)
        assert seen["shape"] == (5, 3, 40)
        dev = par.transform_sharded(fruit, X, block=block, on_device=True)
        assert isinstance(dev, torch.Tensor) and tuple(dev.shape) == full.shape
        np.save(os.path.join(out_dir, f"sharded{rank}.npy"), full)
    finally:
        dist.destroy_process_group()


def test_transform_sharded_from_the_root_alone(tmp_path):
    """parallel.transform_sharded, world 2 over gloo: only rank 0 passes the batch (rank 1
    passes None and receives it by broadcast), both slices' word lists are sharded, and every
    rank ends up with the unsharded feature matrix."""
    import fruits_amd as fr
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    X = np.random.default_rng(13).random((5, 3, 40))
    blocks = []
    for w in (3, 2):
        words = [str(x) for x in fr.words.of_weight(w, dim=3)]
        depths = orc.cache_plan(words)
        blocks.append(_oracle_block(X, words, depths, list(range(len(words))), 0.0).numpy())
    ref = np.concatenate(blocks, axis=1)
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"sharded{r}.npy")), ref)


def test_gather_index_is_one_map():
    ent = G.manifest["words"]["4,2"]
    parts = par.shard_words(ent["words"], ent["plan"], 4)
    maps = par.column_map(parts, ent["plan"], 3)
    rank_of, pos_of = par.gather_index(maps, 3 * ent["K"])
    for r, m in enumerate(maps):
        assert (rank_of[m] == r).all() and (pos_of[m] == np.arange(len(m))).all()
    with pytest.raises(ValueError):
        par.gather_index(maps[:-1], 3 * ent["K"])
