#!/usr/bin/env python3
"""Benchmark of the ISS hot path on MI355X (the metric of BASELINE.json).

One "step" = one pass of the hot path over one synthetic batch: the iterated
sums of ``of_weight(2, dim=3)`` in EXTENDED mode (W = 15 words, K = 18 sums) of a
``(2048, 3, 1024)`` float64 batch that is already resident in HBM, written as the
reference's ``(K, N, T)`` tensor (BASELINE.json configs[1]).  Prints ONE JSON
line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU.
Two things are then measured:

* the headline ``value``: every rank processes its own batch of the workload above
  (the path is independent per series: weak scaling, no data-path collective), the
  timed region is bracketed by barrier + synchronize, the slowest rank counts;
* ``word_sharded_config4``: north_star's multi-GPU split - BASELINE configs[3], the
  ``fruit_general`` word set ``of_weight(6, 2)`` + ``Indices`` on ``(8192, 3, 1024)``
  with the WORD LIST sharded over the ranks (``fruits_amd.parallel``): every rank
  holds the whole batch, computes the feature columns of its sub-tries in one fused
  launch, and one RCCL all-gather (+ a column permutation) assembles the reference's
  ``(N, F)`` feature matrix on every rank.  Reported: slowest rank's launch, the
  all-gather, bytes gathered, end to end, and a check against the unsharded
  transform on rank 0.

``FRUITS_BENCH_BACKEND=gloo`` rehearses the N > 1 path on a box with fewer GPUs than
ranks (ranks share devices, the collective is staged through the host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SERIES, N_DIMS, N_STEPS_T = 2048, 3, 1024


# --------------------------------------------------------------------------- CPU baselines
def cpu_baseline(words, seconds_budget: float = 12.0):
    """The C oracle (oracle/iss_oracle.c, the reference algorithm restated) on
    the host cores, on a bounded sample of the same workload; plus the literal
    single-process numpy path (oracle/ref_numpy.py, SURVEY.md 8d form 1)."""
    from oracle import c_oracle as corc
    from oracle import ref_numpy as norc
    # the GPU box gives one GPU a 16-core share; never oversubscribe it
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16, corc.num_threads()))
    rng = np.random.default_rng(0)
    n_sample = 256
    X = rng.standard_normal((n_sample, N_DIMS, N_STEPS_T))
    strs = [str(w) for w in words]
    corc.iss_transform(X[:8], strs, "EXTENDED", nthreads=threads)  # warm up / build
    reps, t_total = 0, 0.0
    out = None
    while t_total < seconds_budget and reps < 50:
        t0 = time.perf_counter()
        out = corc.iss_transform(X, strs, "EXTENDED", out=out, nthreads=threads)
        t_total += time.perf_counter() - t0
        reps += 1
    K = out.shape[0]
    value = n_sample * K * N_STEPS_T * reps / t_total
    # single-process numpy (vectorised over the series, one thread)
    n_np = 128
    norc.iss_transform(X[:8], strs, "EXTENDED")
    reps_np, t_np = 0, 0.0
    while t_np < 5.0 and reps_np < 20:
        t0 = time.perf_counter()
        norc.iss_transform(X[:n_np], strs, "EXTENDED")
        t_np += time.perf_counter() - t0
        reps_np += 1
    return {
        "value": value, "unit": "iterated-sum elements/s", "cores": threads, "kind": "port",
        "sample": f"{n_sample} of {N_SERIES} series x {reps} reps, oracle/iss_oracle.c "
                  f"(OpenMP over series, {threads} threads)",
        "numpy_single_process": {
            "value": n_np * K * N_STEPS_T * reps_np / t_np, "unit": "iterated-sum elements/s",
            "cores": 1, "kind": "port",
            "sample": f"{n_np} of {N_SERIES} series x {reps_np} reps, oracle/ref_numpy.py "
                      f"(numpy {np.__version__}, vectorised over series, 1 process)"},
    }


# --------------------------------------------------------------------------- timing helpers
def _event_time_us(torch, fn, reps=20):
    # steady state: at least 20 ms of the launch itself first (a measurement that follows seconds
    # of host work - a hipRTC compilation, a fit - otherwise starts on an idle chip's clocks:
    # the 48-word launch read 172 us after a cold compilation and 152 us after a cached one)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def _batch_stats_us(torch, fn, batch=10, batches=50):
    """Per-launch time of `batches` batches of `batch` back-to-back launches, each batch
    between one HIP event pair on the launch stream (an event pair per launch would put
    two barrier packets between consecutive kernels and inflate them by ~10 %)."""
    ts = []
    for _ in range(batches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / batch * 1e3)
    ts = np.asarray(ts)
    return {"median_us": float(np.median(ts)), "min_us": float(ts.min()), "max_us": float(ts.max()),
            "batches": int(batches), "launches_per_batch": int(batch)}


# --------------------------------------------------------------------------- pipelines
def _device_batch(torch, shape, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn(shape, dtype=torch.float64, device="cuda", generator=g)


class _Pipeline:
    """INC -> ISS(words, EXTENDED, weighting) -> sieves on a device-resident RAW batch, fitted on
    the first `n_fit` series: ONE fused launch (all words, or a rank's share of them) that forms
    the increments while it stages the raw rows - what FruitSlice.transform runs."""

    def __init__(self, torch, fr, nat, shape, words, weighting, sieves, n_fit, seed=0, Xd=None,
                 fit_on_root=None):
        """``fit_on_root`` = (rank, world): the fruit is fitted on rank 0 only and its fitted
        state broadcast (fruits_amd.parallel.fit_on_root) instead of fitted on every rank."""
        self.torch, self.fr, self.nat = torch, fr, nat
        self.N, self.D, self.T = shape
        self.Xd = _device_batch(torch, shape, seed) if Xd is None else Xd
        fruit = fr.Fruit("bench")
        fruit.add(fr.preparation.INC)
        self.iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=weighting)
        fruit.add(self.iss)
        fruit.add(*sieves)
        self.slc = fruit.get_slice()
        self.slc.fit_sample_size = 1.0
        np.random.seed(0)
        if fit_on_root is None:
            fruit.fit(self.Xd[:n_fit].cpu().numpy())
        else:
            from fruits_amd import parallel as par
            rank, world = fit_on_root
            par.fit_on_root(fruit, self.Xd[:n_fit].cpu().numpy() if rank == 0 else None, rank, world)
            self.slc = fruit.get_slice()
            self.iss = self.slc.get_iss()[0]
        self.fruit = fruit
        self.cache = fr.cache.SharedSeedCache(None)
        self.cache.adopt_device_input(self.Xd)
        self.slc._attach(self.cache)
        self.chain = self.slc._fusable_preparation(self.T)
        assert self.chain is not None, "the bench pipelines fuse their preparation"
        # (Indices do not look at the data, L1 measures the cache's raw input)
        self.lk = self.iss.lookup_device(self.Xd)
        self.strings = [str(w) for w in self.iss.words]
        self.depths = [self.iss._depth(i) for i in range(len(self.strings))]
        self.per_sum = sum(s.nfeatures() for s in self.slc.get_sieves())

    def launch(self, indices=None):
        """(fn, feats, pipe): fn enqueues the fused launch of the given words on the raw batch."""
        torch, nat = self.torch, self.nat
        pipe = self.slc._fused(self.T, indices=indices)
        assert pipe is not None, "the bench pipelines are inside the fused set"
        assert pipe.set_preparation(self.D, *self.chain), "INC is formed while the rows are staged"
        feats = torch.empty((self.N, pipe.n_features), dtype=torch.float64, device="cuda")
        rows = 0 if self.lk is None else int(self.lk.shape[0])
        wb = int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, self.N, rows))
        work = torch.empty(max(wb, 1), dtype=torch.uint8, device="cuda")
        pipe.prepare(self.N)
        return (lambda: pipe.run(self.Xd, self.lk, feats=feats, work=work)), feats, pipe

    def figures(self, pipe, t_us):
        N, T = self.N, self.T
        K, d_used = pipe.plan.rows, pipe.plan.dims_used
        lookup_b = 0 if self.lk is None else 8.0 * self.lk.numel()
        return {
            "launch_us": t_us, "K": K, "features": pipe.n_features, "nodes": pipe.plan.nodes,
            "kernel": ("fused walk compiled for this pipeline at prepare time (hipRTC: the sieves "
                       "as immediates)" if pipe.jit_loaded() else "fused walk, generic instance"),
            "elements_per_s": N * K * T / (t_us * 1e-6),
            "algorithmic_bytes": 8.0 * N * T * d_used + lookup_b + 8.0 * N * pipe.n_features,
            "equivalent_materialised_GBs": (8.0 * N * T * (d_used + K) + lookup_b) / (t_us * 1e-6) / 1e9,
            "note": "the raw batch goes in, INC is formed in the staging; no (K,N,T) tensor is "
                    "written; the GB/s figure is what a materialising run of the same work would "
                    "have needed, not achieved bandwidth",
        }


def _config4(torch, fr, nat, Xd=None, fit_on_root=None):
    return _Pipeline(torch, fr, nat, (8192, 3, 1024), fr.words.of_weight(6, dim=2),
                     fr.iss.weighting.Indices(), [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END],
                     n_fit=128, Xd=Xd, fit_on_root=fit_on_root)


def _broadcast_batch(torch, dist, shape, rank, backend):
    """The batch generated ONCE, on rank 0, and broadcast device to device (RCCL; a gloo
    rehearsal stages it through the host) - instead of every rank generating / uploading it."""
    t0 = time.perf_counter()
    Xd = (_device_batch(torch, shape, 0) if rank == 0
          else torch.empty(shape, dtype=torch.float64, device="cuda"))
    if backend == "nccl":
        dist.broadcast(Xd, src=0)
    else:
        host = Xd.cpu()
        dist.broadcast(host, src=0)
        Xd.copy_(host)
    torch.cuda.synchronize()
    return Xd, (time.perf_counter() - t0) * 1e3


def sweep(torch, fr, nat, dev, copy_GBs, budget_s=25.0):
    """The materialising walk of of_weight(2,3) EXTENDED (K = 18) over N x T, once with the
    pre-compiled static program and once through the record interpreter (FRUITS_HIP_STATIC=0),
    and of_weight(4,2) EXTENDED (K = 115, no static program) at the headline shape; every cell
    with its fraction of the 8 TB/s spec and of the copy rate measured on this box.  Cells are
    dropped when the time budget is spent (largest first kept: the order below)."""
    t_start = time.perf_counter()
    cells = []
    w2 = fr.words.of_weight(2, dim=N_DIMS)

    def cell(words, N, T, static):
        plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
        K = plan.rows
        if 8.0 * N * T * (plan.dims_used + K) > 40e9:
            return None
        prev = os.environ.get("FRUITS_HIP_STATIC")
        os.environ["FRUITS_HIP_STATIC"] = "1" if static else "0"
        try:
            Xs = _device_batch(torch, (N, N_DIMS, T), 1)
            buf = torch.empty((K, N, T), dtype=torch.float64, device=dev)
            if static:
                plan.prepare(N, T)
            t = _event_time_us(torch, lambda: plan.run(Xs, None, out=buf), reps=10)
        finally:
            if prev is None:
                os.environ.pop("FRUITS_HIP_STATIC", None)
            else:
                os.environ["FRUITS_HIP_STATIC"] = prev
        b_alg = 8.0 * N * T * (plan.dims_used + K)
        gbs = b_alg / (t * 1e-6) / 1e9
        del Xs, buf
        return {"words": f"of_weight({len(words)} words)", "K": K, "N": N, "T": T,
                "path": "static program" if static else "interpreter", "kernel_us": t,
                "GBs": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS, "frac_of_on_box_copy": gbs / copy_GBs}
    order = [(w2, 2048, 1024), (w2, 8192, 1024), (w2, 512, 1024), (w2, 1024, 1024), (w2, 3072, 1024),
             (w2, 4096, 1024), (w2, 16384, 1024), (w2, 2048, 4096), (w2, 8192, 256), (w2, 2048, 256),
             (w2, 512, 4096), (w2, 8192, 4096), (w2, 512, 256), (w2, 16384, 256)]
    for words, N, T in order:
        for static in (True, False):
            if time.perf_counter() - t_start > budget_s:
                break
            c = cell(words, N, T, static)
            if c is not None:
                cells.append(c)
        torch.cuda.empty_cache()
    if time.perf_counter() - t_start <= budget_s + 5.0:
        c = cell(fr.words.of_weight(4, dim=2), N_SERIES, N_STEPS_T, True)
        if c is not None:
            c["path"] = "lean materialising walk (115 nodes: no static program; DESIGN.md 4.1c)"
            cells.append(c)
    return {"cells": cells, "on_box_copy_GBs": copy_GBs,
            "note": "static programs cover T in (512, 1024]; other lengths run the interpreter "
                    "(or the wave-per-series kernels for T <= 384) on both lines; 'interpreter' = "
                    "no static program: the record interpreter, and from two resident rounds of "
                    "workgroups on the lean materialising walk (DESIGN.md 4.1c)"}


def extras(torch, fr, nat, dev, quick=False):
    """Secondary measurements on the same GPU (not the headline value): the metric's
    "48 words" reading, what the box streams, and the fused pipelines of BASELINE
    configs[2], [3] and [4] on ONE GPU."""
    out = {}
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N_SERIES, N_DIMS, N_STEPS_T))
    Xd = nat.to_device(X)
    # (a) words[i % 15] tiled to 48, SINGLE mode: K = 48 rows, 855.6 MB algorithmic
    w15 = fr.words.of_weight(2, dim=N_DIMS)
    w48 = [w15[i % 15] for i in range(48)]
    plan = fr.ISS(w48)._plan(0, 48)
    # (not one of the pre-compiled word sets: fr_plan_prepare compiles its static program with
    # hipRTC - once, cached on disk - as a caller that launches a plan repeatedly would)
    plan.prepare(N_SERIES, N_STEPS_T)
    # (the time of this launch depends on where the driver places its 805 MB output: the same
    # kernel in the same process reads 150 us on one allocation and 166 us on the next, stable
    # within an allocation - three placements, the median is the figure)
    placements, shift = [], []
    for trial in range(3):
        buf = torch.empty((48, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
        placements.append(_event_time_us(torch, lambda: plan.run(Xd, None, out=buf)))
        del buf
        torch.cuda.empty_cache()
        shift.append(torch.empty((trial + 1) * 37_000_001, dtype=torch.uint8, device=dev))
    del shift
    torch.cuda.empty_cache()
    t = float(np.median(placements))
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (3 + 48)
    out["words48_single"] = {"kernel_us": t, "elements_per_s": N_SERIES * 48 * N_STEPS_T / (t * 1e-6),
                             "GBs": b_alg / (t * 1e-6) / 1e9, "frac": b_alg / (t * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "placements_us": placements,
                             "best_placement_frac": b_alg / (min(placements) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "static_programs_compiled_at_run_time": plan.jit_loaded()}
    # (a') what this box sustains (SURVEY.md 8d asks for an on-box peak next to the 8 TB/s
    # spec): a device fill and a device-to-device copy of the size of the output tensor
    big = torch.empty((18, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
    big2 = torch.empty_like(big)
    t_fill = _event_time_us(torch, lambda: big.fill_(1.0))
    t_copy = _event_time_us(torch, lambda: big2.copy_(big))
    nbytes = big.numel() * 8
    out["on_box_stream"] = {"fill_GBs": nbytes / (t_fill * 1e-6) / 1e9,
                            "copy_read_plus_write_GBs": 2 * nbytes / (t_copy * 1e-6) / 1e9,
                            "bytes": nbytes,
                            "note": "torch fill_ / copy_ of a buffer of the (K,N,T) tensor's size"}
    del big, big2
    # (b) config 3 shape: INC -> ISS(of_weight(4,2) EXTENDED, Indices) -> NPI(q=(.5,1)), END: one
    # fused launch on the raw batch
    p3 = _Pipeline(torch, fr, nat, (N_SERIES, N_DIMS, N_STEPS_T), fr.words.of_weight(4, dim=2),
                   fr.iss.weighting.Indices(), [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END],
                   n_fit=128, Xd=Xd)
    fn, _, pipe3 = p3.launch()
    t = _event_time_us(torch, fn)
    p3.fruit.transform(X)   # first call builds plans / uploads tables
    t0 = time.perf_counter()
    p3.fruit.transform(X)
    e2e = time.perf_counter() - t0
    out["config3_fused_pipeline"] = dict(
        p3.figures(pipe3, t), fruit_transform_end_to_end_ms=e2e * 1e3,
        workload="BASELINE configs[2] shape: of_weight(4,2) EXTENDED + Indices, (2048,3,1024), "
                 "INC -> ISS -> NPI(q=(0.5,1)), END, one fused launch")
    del p3, fn, pipe3
    # (b') the shape sweep: what the materialising walk reaches away from the headline shape
    out["sweep"] = sweep(torch, fr, nat, dev, out["on_box_stream"]["copy_read_plus_write_GBs"],
                         budget_s=4.0 if quick else 25.0)
    del Xd
    if quick:
        return out
    # (c) config 4 on ONE GPU: fruit_general's of_weight(6,2) + Indices, (8192,3,1024),
    # INC -> ISS -> NPI(q=(.5,1)), END; K = 1351 (the reference's tensor would be 90 GB)
    p4 = _config4(torch, fr, nat)
    fn, _, pipe4 = p4.launch()
    out["config4_single_gpu"] = dict(
        p4.figures(pipe4, _event_time_us(torch, fn, reps=5)),
        workload="BASELINE configs[3] on one GPU: of_weight(6,2) EXTENDED + Indices, raw "
                 "(8192,3,1024) in, INC -> ISS -> NPI(q=(0.5,1)), END, one fused launch")
    del p4, fn, pipe4
    torch.cuda.empty_cache()
    # (d) config 5 on ONE GPU: fruit_twi slice 1, of_weight(9,1) + L1, (8192,6,4096)
    p5 = _Pipeline(torch, fr, nat, (8192, 6, 4096), fr.words.of_weight(9, dim=1),
                   fr.iss.weighting.L1(), [fr.sieving.NPI, fr.sieving.END], n_fit=32)
    fn, _, pipe5 = p5.launch()
    out["config5_single_gpu"] = dict(
        p5.figures(pipe5, _event_time_us(torch, fn, reps=3)),
        workload="BASELINE configs[4] on one GPU: of_weight(9,1) EXTENDED + L1, (8192,6,4096) "
                 "(N chosen: SURVEY.md 0.3), INC -> ISS -> NPI, END, one fused launch over 4 "
                 "time chunks")
    del p5, fn, pipe5
    torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------- word-sharded config 4
def word_sharded_config4(torch, fr, nat, dist, rank, world, backend):
    """BASELINE configs[3]: the word list of fruit_general's first slice sharded over the
    ranks, features all-gathered (fruits_amd.parallel).  Every rank holds the same batch."""
    from fruits_amd import parallel as par
    Xd, bcast_ms = _broadcast_batch(torch, dist, (8192, 3, 1024), rank, backend)
    t0 = time.perf_counter()
    p = _config4(torch, fr, nat, Xd=Xd, fit_on_root=(rank, world))
    fit_ms = (time.perf_counter() - t0) * 1e3
    N, T = p.N, p.T
    parts = par.shard_words(p.strings, p.depths, world)
    maps = par.column_map(parts, p.depths, p.per_sum)
    n_features = p.slc.nfeatures()
    fn, local, pipe = p.launch(indices=parts[rank])
    # (1) the rank's fused launch alone
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    my_ms = float(np.median(ts))
    # (2) the all-gather alone and (3) launch + gather + permutation, device resident
    par.gather_features(local, maps, n_features, rank, world)   # warm-up (RCCL channel set-up)
    gather_ms, e2e_ms = [], []
    full = None
    for _ in range(5):
        dist.barrier()
        torch.cuda.synchronize()
        tm = {}
        par.gather_features(local, maps, n_features, rank, world, timings=tm)
        gather_ms.append(tm.get("allgather_s", 0.0) * 1e3)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        full = par.gather_features(local, maps, n_features, rank, world)
        torch.cuda.synchronize()
        e2e_ms.append((time.perf_counter() - t0) * 1e3)
    mine = {"rank": rank, "launch_ms": my_ms, "allgather_ms": float(np.median(gather_ms)),
            "end_to_end_ms": float(np.median(e2e_ms)), "K": pipe.plan.rows,
            "features": pipe.n_features, "nodes": pipe.plan.nodes}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    series_leg = series_sharded_config4(torch, dist, rank, world, backend, p)
    if rank != 0:
        return None
    # (4) rank 0: the unsharded transform of the same batch, every column compared
    fn1, feats1, pipe1 = p.launch()
    t_one = _event_time_us(torch, fn1, reps=3) / 1e3
    fn1()
    torch.cuda.synchronize()
    diff = (full - feats1).abs()
    cols_ok = int((diff.max(dim=0).values == 0).sum().item())
    launches = [e["launch_ms"] for e in everyone]
    e2e = max(e["end_to_end_ms"] for e in everyone)
    width = max(len(m) for m in maps)
    K_total = pipe1.plan.rows
    return {
        "workload": "BASELINE configs[3]: fruit_general slice-1 word set of_weight(6,2) EXTENDED + "
                    "Indices, (8192,3,1024) float64 on EVERY rank, INC -> ISS -> NPI(q=(0.5,1)), END; "
                    "word list sharded over the ranks by sub-trie, one fused launch per rank, one "
                    "padded all_gather_into_tensor of the (N, F_r) blocks + column permutation",
        "backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else " (host-staged rehearsal)"),
        "world_size": dist.get_world_size(), "N": N, "D": p.D, "T": T, "words": len(p.strings),
        "K": K_total, "features": n_features,
        "rank_launch_ms": launches, "slowest_launch_ms": max(launches),
        "balance": float(np.mean(launches) / max(launches)),
        "K_per_rank": [e["K"] for e in everyone], "nodes_per_rank": [e["nodes"] for e in everyone],
        "allgather_ms": max(e["allgather_ms"] for e in everyone),
        "gathered_bytes_per_rank": int(N * width * 8),
        "gathered_bytes_total": int(world * N * width * 8),
        "end_to_end_ms": e2e,
        "elements_per_s": N * K_total * T / (e2e * 1e-3),
        "single_rank_unsharded_launch_ms": t_one,
        "speedup_vs_single_rank_launch": t_one / e2e,
        "equals_unsharded_transform": bool(cols_ok == n_features),
        "columns_checked": n_features, "columns_bit_identical": cols_ok,
        "max_abs_diff": float(diff.max().item()),
        "batch": f"generated on rank 0, broadcast to the ranks ({bcast_ms:.1f} ms incl. generation)",
        "fit": f"on rank 0 only, fitted state broadcast (pipeline set-up {fit_ms:.1f} ms on rank 0)",
        "series_sharded_config4": None if series_leg is None else dict(
            series_leg, single_rank_unsharded_launch_ms=t_one,
            equals_unsharded_transform=bool((series_leg.pop("_full") == feats1).all().item())),
    }


def series_sharded_config4(torch, dist, rank, world, backend, p):
    """The same workload with the SERIES sharded instead: every rank runs the whole word list
    (one fused launch) on its N / world rows of the batch - no collective on the data path."""
    from fruits_amd import parallel as par
    nat = p.nat
    rows = par.shard_series(p.N, rank, world)
    Xr = p.Xd[rows].contiguous()
    n_r = int(Xr.shape[0])
    pipe = p.slc._fused(p.T)
    assert pipe.set_preparation(p.D, *p.chain) and p.lk.shape[0] == 1
    feats = torch.empty((n_r, pipe.n_features), dtype=torch.float64, device="cuda")
    wb = int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, n_r, 1))
    work = torch.empty(max(wb, 1), dtype=torch.uint8, device="cuda")
    pipe.prepare(n_r)
    fn = lambda: pipe.run(Xr, p.lk, feats=feats, work=work)
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    everyone = [None] * world
    dist.all_gather_object(everyone, {"rank": rank, "rows": n_r, "launch_ms": float(np.median(ts))})
    # for the check only (not part of the path): the row blocks assembled on rank 0
    tallest = max(e["rows"] for e in everyone)
    padded = torch.zeros((tallest, pipe.n_features), dtype=torch.float64, device="cuda")
    padded[:n_r] = feats
    if backend == "nccl":
        flat = torch.empty((world * tallest, pipe.n_features), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(flat, padded)
    else:
        flat_h = torch.empty((world * tallest, pipe.n_features), dtype=torch.float64)
        dist.all_gather_into_tensor(flat_h, padded.cpu())
        flat = flat_h.cuda()
    if rank != 0:
        return None
    blocks = flat.view(world, tallest, pipe.n_features)
    full = torch.cat([blocks[r, :everyone[r]["rows"]] for r in range(world)], dim=0)
    launches = [e["launch_ms"] for e in everyone]
    return {
        "workload": "BASELINE configs[3] with the series sharded: every rank the whole word list "
                    "(K = 1351) on N / world rows, one fused launch, no data-path collective",
        "rows_per_rank": [e["rows"] for e in everyone], "rank_launch_ms": launches,
        "slowest_launch_ms": max(launches), "balance": float(np.mean(launches) / max(launches)),
        "elements_per_s": p.N * pipe.plan.rows * p.T / (max(launches) * 1e-3),
        "_full": full,
    }


# --------------------------------------------------------------------------- main
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--quick-extras", action="store_true", help="skip the config 4 / 5 extras")
    ap.add_argument("--no-word-shard", action="store_true")
    ap.add_argument("--groups", type=int, default=0)
    args = ap.parse_args()

    import torch
    import fruits_amd as fr
    from fruits_amd import _native as nat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FRUITS_BENCH_BACKEND", "nccl")
    distributed = world > 1 or bool(os.environ.get("FRUITS_BENCH_FORCE_DIST"))
    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(device_index)
    dist = None
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version
    # banner when its first communicator comes up, gloo its connection lines): everything but
    # the result goes to stderr - file descriptor 1 is pointed at 2 until the line is written.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    dev = nat.require_device()

    words = fr.words.of_weight(2, dim=N_DIMS)
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
    plan = iss._plan(0, len(words))
    K = plan.rows
    rng = np.random.default_rng(rank)
    X = rng.standard_normal((N_SERIES, N_DIMS, N_STEPS_T))
    Xd = nat.to_device(X)
    out = torch.empty((K, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
    plan.prepare(N_SERIES, N_STEPS_T, args.groups)

    def step():
        plan.run(Xd, None, out=out, layout="KNT", groups=args.groups)

    def barrier():
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(args.warmup, 10)):   # >= 10 warm-ups whatever the driver asks for
        step()
    torch.cuda.synchronize()
    barrier()
    # Timed region: exactly `steps` back-to-back launches between barrier + synchronize.
    # Kernel duration: ONE pair of HIP events around those launches on the stream the
    # kernel is launched on (torch's current stream is passed through the C ABI);
    # average = elapsed / steps, an upper bound that still contains the launch gaps.
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record()
    for _ in range(args.steps):
        step()
    ev_b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_avg_s = ev_a.elapsed_time(ev_b) / 1e3 / args.steps

    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the timed kernel really produced the tensor (one row against numpy)
    row0 = out[0, :4].cpu().numpy()
    assert np.allclose(row0, np.cumsum(X[:4, 0] ** 2, axis=1), rtol=1e-9)

    # distribution of the per-launch time: median / min / max over >= 50 event-timed batches
    batches = _batch_stats_us(torch, step) if rank == 0 else None

    elements = N_SERIES * K * N_STEPS_T
    value = elements * args.steps * world / elapsed
    d_used = plan.dims_used
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (d_used + K)  # read X once, write every sum once
    achieved = b_alg / kernel_avg_s / 1e9
    traffic, traffic_src = None, None
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                tj = json.load(f)
            traffic = tj.get("iss_walk_bytes_per_launch")
            traffic_src = ("constant from profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / "
                           "WRITE_SIZE passes over this command, " + str(tj.get("round", "r01")) +
                           "; FETCH_SIZE x2 per MI355X_MICROARCH.md) - not measured in this run")
        except Exception:
            traffic = None
    res = {
        "metric": "ISS features/sec (N*words*T/s), (2048,3,1024) weight-2 words",
        "value": value,
        "unit": "iterated-sum elements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: fruits.words.of_weight(2, dim=3) EXTENDED "
                        "(W=15 words, K=18 iterated sums), float64 (2048,3,1024) per GPU, "
                        "(K,N,T) tensor materialised in HBM",
            "N": N_SERIES, "D": N_DIMS, "T": N_STEPS_T, "words": len(words), "K": K,
            "sharding": "series (one batch per GPU), no data-path collective; the word-sharded "
                        "RCCL all-gather split of configs[3] is measured beside it for n_gpus > 1 "
                        "(word_sharded_config4)",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "iss_walk_static_kernel (of_weight(2,3) program, one group per series)", "algorithmic_bytes_per_launch": b_alg,
            "kernel_avg_us": kernel_avg_s * 1e6,
            "timing": "one HIP event pair around the `steps` launches of the timed region / steps",
            "batches": batches,
        },
    }
    del out
    if distributed and not args.no_word_shard:
        ws = word_sharded_config4(torch, fr, nat, dist, rank, world, backend)
        if rank == 0:
            res["word_sharded_config4"] = ws
    if rank == 0 and world == 1 and not args.no_extras:
        res["extras"] = extras(torch, fr, nat, dev, quick=args.quick_extras)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(words)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
