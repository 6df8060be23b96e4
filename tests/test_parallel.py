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
