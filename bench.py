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
* ``word_sharded_config4``: north_star's multi-GPU split through the PRODUCT entry
  ``fruits_amd.parallel.transform_sharded`` - BASELINE configs[3], the ``fruit_general``
  word set ``of_weight(6, 2)`` + ``Indices`` on ``(8192, 3, 1024)``: the batch exists on rank 0
  only and is broadcast device to device, the fruit is fitted on rank 0, every rank computes
  the feature columns of its sub-tries in one fused launch, one RCCL all-gather and one column
  gather assemble the reference's ``(N, F)`` on every rank; checked against the unsharded
  transform on rank 0.

N = 1 adds, INSIDE the ``roofline`` object (the driver's record keeps it verbatim):
``secondary`` - the metric's 48-word reading, of_weight(4,2) materialised (K = 115), what the box
copies, and the fused pipelines of configs[2..4] on one GPU (us, fraction; for the fused ones the
VALU issue fraction of the last committed profile) - and ``sweep_N_T_static_interp``; ``extras``
holds ``cold_start`` (an empty user cache of run-time compiled kernels).

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


def _isolated_time_us(torch, fn, n=30):
    """Median duration of ISOLATED launches: every launch between its own HIP event pair, the device
    drained in front of it (what `rocprofv3 --kernel-trace`, which serialises dispatches, sees; an
    event pair adds a few us of barrier packets).  Beside the back-to-back figure it separates a
    kernel's own time from how consecutive launches overlap on the same output."""
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


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


def sweep(torch, fr, nat, dev, budget_s=25.0):
    """The materialising walk of of_weight(2,3) EXTENDED (K = 18) over N x T, once with the
    pre-compiled static program and once without (FRUITS_HIP_STATIC=0: the record interpreter /
    the lean materialising walk): rows [N, T, fraction of 8 TB/s with, without].  Cells are
    dropped when the time budget is spent."""
    t_start = time.perf_counter()
    w2 = fr.words.of_weight(2, dim=N_DIMS)
    plan = fr.ISS(w2, mode=fr.ISSMode.EXTENDED)._plan(0, len(w2))
    K = plan.rows

    def frac(N, T, static):
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
        del Xs, buf
        return round(8.0 * N * T * (plan.dims_used + K) / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)
    rows = []
    order = [(2048, 1024), (8192, 1024), (512, 1024), (1024, 1024), (3072, 1024), (4096, 1024),
             (16384, 1024), (2048, 4096), (8192, 256), (2048, 256), (512, 4096), (8192, 4096),
             (512, 256), (16384, 256)]
    for N, T in order:
        if time.perf_counter() - t_start > budget_s:
            break
        rows.append([N, T, frac(N, T, True), frac(N, T, False)])
        torch.cuda.empty_cache()
    return rows


def _issue_fractions():
    """VALU issue fractions of the fused launches from the last committed profile run
    (profiles/fused_issue.json, written from rocprofv3 SQ counters: not measured in this run)."""
    try:
        with open(os.path.join(ROOT, "profiles", "fused_issue.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def _fused_entry(p, pipe, t_us, issue, key):
    """A fused pipeline's launch for roofline.secondary: these launches write (N, F) only, so the
    bound that applies is vector issue (profiles/), not HBM; `equiv_frac` = what a materialising
    launch of the same sums would have to stream, as a fraction of 8 TB/s (labelled: above 1 by
    construction - avoided traffic, not bandwidth)."""
    K, d_used = pipe.plan.rows, pipe.plan.dims_used
    equiv = 8.0 * p.N * p.T * (d_used + K) / (t_us * 1e-6) / 1e9
    e = {"us": round(t_us, 1), "K": K, "nodes": pipe.plan.nodes,
         "elements_per_s": float(f"{p.N * K * p.T / (t_us * 1e-6):.4g}"),
         "equiv_frac": round(equiv / HBM_PEAK_GBS, 3),
         "kernel": ("pieces" if pipe.pieces_loaded() else
                    "straight-line plan" if pipe.jit_loaded(static_only=True) else
                    "own sieves" if pipe.jit_loaded() else "generic")}
    if key in issue:
        e["valu_issue_frac"] = issue[key]
    return e


def _bench_fruit(fr, words, weighting, sieves):
    fruit = fr.Fruit("bench")
    fruit.add(fr.preparation.INC)
    fruit.add(fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=weighting))
    fruit.add(*sieves)
    fruit.get_slice().fit_sample_size = 1.0
    return fruit


def cold_start(torch, fr, nat, which, bundle=True):
    """What a one-shot user sees on a machine that has never run this fruit: an EMPTY user
    cache of run-time compiled kernels (FRUITS_HIP_JIT_CACHE = a fresh directory; the kernels
    shipped with the build, fruits_amd/jit_bundle, are still there unless ``bundle`` is False -
    the case of a fruit the build ships nothing for).  ms of the first
    Fruit.transform, of the transforms while a compilation runs in the background, seconds until
    the compiled kernels take over, ms after."""
    import tempfile
    prev = {k: os.environ.get(k) for k in ("FRUITS_HIP_JIT_CACHE", "FRUITS_AMD_AUTO_PREPARE", "FRUITS_HIP_JIT_BUNDLE")}
    os.environ["FRUITS_HIP_JIT_CACHE"] = tempfile.mkdtemp(prefix="fruits_cold_")
    os.environ["FRUITS_AMD_AUTO_PREPARE"] = "1"
    if not bundle:       # (a fruit the build does not ship kernels for: everything through the compiler)
        os.environ["FRUITS_HIP_JIT_BUNDLE"] = ""
    try:
        shape, words = {"cfg3": ((2048, 3, 1024), fr.words.of_weight(4, dim=2)),
                        "cfg4": ((8192, 3, 1024), fr.words.of_weight(6, dim=2))}[which]
        X = np.random.default_rng(0).standard_normal(shape)
        fruit = _bench_fruit(fr, words, fr.iss.weighting.Indices(),
                             [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END])
        np.random.seed(0)
        fruit.fit(X[:128])

        def once():
            t0 = time.perf_counter()
            fruit.transform(X)
            return (time.perf_counter() - t0) * 1e3
        first = once()
        pipe = fruit.get_slice()._fused(shape[2])

        def compiled():
            return pipe.pieces_loaded() > 0 or pipe.jit_loaded(static_only=True) > 0
        on_first = compiled()
        during, t_start = [], time.perf_counter()
        while not compiled() and time.perf_counter() - t_start < 90.0:
            during.append(once())
        takeover = time.perf_counter() - t_start
        pending = getattr(pipe, "_pending", None)
        if pending is not None:
            pending.result(timeout=120)
        after = sorted(once() for _ in range(3))[1]
        return {"first_transform_ms": round(first, 1), "compiled_kernels_on_first_launch": bool(on_first),
                "transform_ms_while_compiling": round(float(np.median(during)), 1) if during else None,
                "seconds_until_compiled": round(takeover, 1) if not on_first else 0.0,
                "transform_ms_after": round(after, 1), "took_over": bool(compiled()),
                "note": "Fruit.transform end to end (upload of X, launch, download of the features)"}
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def extras(torch, fr, nat, dev, quick=False):
    """Secondary measurements on the same GPU (not the headline value).  Returns (secondary,
    sweep, details): `secondary` and `sweep` go INTO the roofline object (the driver's record
    keeps that verbatim), the details into `extras`."""
    sec, det = {}, {}
    issue = _issue_fractions()
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N_SERIES, N_DIMS, N_STEPS_T))
    Xd = nat.to_device(X)
    # (a) the metric's "48 words": words[i % 15] tiled to 48, SINGLE mode, K = 48, 855.6 MB
    w15 = fr.words.of_weight(2, dim=N_DIMS)
    plan = fr.ISS([w15[i % 15] for i in range(48)])._plan(0, 48)
    plan.prepare(N_SERIES, N_STEPS_T)     # (its static program: hipRTC, cached on disk)
    placements, isolated, shift = [], [], []
    for trial in range(3):                # (the time depends on where the 805 MB output lies)
        buf = torch.empty((48, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
        placements.append(_event_time_us(torch, lambda: plan.run(Xd, None, out=buf)))
        isolated.append(_isolated_time_us(torch, lambda: plan.run(Xd, None, out=buf)))
        del buf
        torch.cuda.empty_cache()
        shift.append(torch.empty((trial + 1) * 37_000_001, dtype=torch.uint8, device=dev))
    del shift
    torch.cuda.empty_cache()
    t = float(np.median(placements))
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (3 + 48)
    sec["words48"] = {"us": round(t, 1), "frac": round(b_alg / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                      "placements_us": [round(v, 1) for v in placements],
                      "isolated_launch_us": [round(v, 1) for v in isolated]}
    # (a') of_weight(4,2) EXTENDED, K = 115, materialised (1.9 GB): three placements too
    w42 = fr.words.of_weight(4, dim=2)
    plan = fr.ISS(w42, mode=fr.ISSMode.EXTENDED)._plan(0, len(w42))
    plan.prepare(N_SERIES, N_STEPS_T)
    placements = []
    for trial in range(3):
        buf = torch.empty((plan.rows, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
        placements.append(_event_time_us(torch, lambda: plan.run(Xd, None, out=buf), reps=5))
        del buf
        torch.cuda.empty_cache()
    t = float(np.median(placements))
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (plan.dims_used + plan.rows)
    sec["K115"] = {"us": round(t, 1), "frac": round(b_alg / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                   "placements_us": [round(v, 1) for v in placements]}
    # (a'') what this box streams: fill / copy of a buffer of the headline tensor's size
    big = torch.empty((18, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
    big2 = torch.empty_like(big)
    t_fill = _event_time_us(torch, lambda: big.fill_(1.0))
    t_copy = _event_time_us(torch, lambda: big2.copy_(big))
    nbytes = big.numel() * 8
    copy_GBs = 2 * nbytes / (t_copy * 1e-6) / 1e9
    sec["on_box_copy"] = {"GBs": round(copy_GBs), "frac": round(copy_GBs / HBM_PEAK_GBS, 3),
                          "fill_GBs": round(nbytes / (t_fill * 1e-6) / 1e9)}
    del big, big2
    # (b) config 3: INC -> ISS(of_weight(4,2) EXTENDED, Indices) -> NPI(q=(.5,1)), END, one launch
    p3 = _Pipeline(torch, fr, nat, (N_SERIES, N_DIMS, N_STEPS_T), fr.words.of_weight(4, dim=2),
                   fr.iss.weighting.Indices(), [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END],
                   n_fit=128, Xd=Xd)
    fn, _, pipe3 = p3.launch()
    sec["cfg3"] = _fused_entry(p3, pipe3, _event_time_us(torch, fn), issue, "cfg3")
    del p3, fn, pipe3
    sweep_rows = sweep(torch, fr, nat, dev, budget_s=4.0 if quick else 25.0)
    del Xd
    if quick:
        return sec, sweep_rows, det
    # (c) config 4 on ONE GPU: of_weight(6,2) + Indices, (8192,3,1024); K = 1351
    p4 = _config4(torch, fr, nat)
    fn, _, pipe4 = p4.launch()
    sec["cfg4"] = _fused_entry(p4, pipe4, _event_time_us(torch, fn, reps=5), issue, "cfg4")
    del p4, fn, pipe4
    torch.cuda.empty_cache()
    # (d) config 5 on ONE GPU: of_weight(9,1) + L1, (8192,6,4096), four time chunks
    p5 = _Pipeline(torch, fr, nat, (8192, 6, 4096), fr.words.of_weight(9, dim=1),
                   fr.iss.weighting.L1(), [fr.sieving.NPI, fr.sieving.END], n_fit=32)
    fn, _, pipe5 = p5.launch()
    sec["cfg5"] = _fused_entry(p5, pipe5, _event_time_us(torch, fn, reps=3), issue, "cfg5")
    del p5, fn, pipe5
    torch.cuda.empty_cache()
    # (f) the reference's experiments/fruit_reduced.py (four slices: Reals + Indices, Arctic, two
    # CosWISS; 4431 features) on (2048,1,1024): Fruit.fit (3798 order statistics selected on the
    # device) and Fruit.transform, wall clock with the input on the host
    det["fruit_reduced"] = fruit_reduced_times(torch)
    # (e) a cold machine
    det["cold_start"] = {w: cold_start(torch, fr, nat, w) for w in ("cfg3", "cfg4")}
    det["cold_start"]["cfg4_without_shipped_kernels"] = cold_start(torch, fr, nat, "cfg4", bundle=False)
    torch.cuda.empty_cache()
    return sec, sweep_rows, det


def fruit_reduced_times(torch, reps=5):
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import bench_pipeline as bp
    X = np.random.default_rng(0).standard_normal((2048, 1, 1024)).cumsum(axis=2)
    fruit = bp.build_reduced()
    fits, trs = [], []
    for i in range(reps + 1):
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fruit.fit(X)
        torch.cuda.synchronize(); fits.append(time.perf_counter() - t0)
    fruit.transform(X)
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fruit.transform(X)
        torch.cuda.synchronize(); trs.append(time.perf_counter() - t0)
    return {"shape": [2048, 1, 1024], "features": int(fruit.nfeatures()),
            "fit_ms": round(float(np.median(fits[1:])) * 1e3, 1),
            "transform_ms": round(float(np.median(trs)) * 1e3, 1)}


# --------------------------------------------------------------------------- word-sharded config 4
def word_sharded_config4(torch, fr, nat, dist, rank, world, backend):
    """BASELINE configs[3] through the PRODUCT entry, fruits_amd.parallel.transform_sharded: the
    batch exists on rank 0 only (the others pass None: uploaded once, broadcast device to
    device), the fruit is fitted on rank 0 (state broadcast), the word list of the slice is
    sharded over the ranks by sub-trie, every rank one fused launch on the raw batch, one padded
    all-gather of the (N, F_r) blocks + ONE column gather."""
    from fruits_amd import parallel as par
    shape = (8192, 3, 1024)
    Xd = _device_batch(torch, shape, 0) if rank == 0 else None
    fruit = _bench_fruit(fr, fr.words.of_weight(6, dim=2), fr.iss.weighting.Indices(),
                         [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END])
    np.random.seed(0)
    t0 = time.perf_counter()
    par.fit_on_root(fruit, Xd[:128].cpu().numpy() if rank == 0 else None, rank, world)
    fit_ms = (time.perf_counter() - t0) * 1e3
    # steady state: this rank's shard gets its own kernels now (its plan in pieces: ~10 s of hipRTC
    # on a cold cache, every rank at once) - transform_sharded itself would have them compiled in
    # the background and run the generic instance meanwhile
    slc0 = fruit.get_slice()
    iss0 = slc0.get_iss()[0]
    strings0 = [str(w) for w in iss0.words]
    depths0 = [iss0._depth(i) for i in range(len(strings0))]
    mine0 = par.shard_words(strings0, depths0, world)[rank]
    if mine0:
        slc0._fused(shape[2], indices=mine0).prepare(shape[0])
    par.transform_sharded(fruit, Xd, rank, world, on_device=True)      # (plans, tables, RCCL set-up)
    par.transform_sharded(fruit, Xd, rank, world, on_device=True)
    runs, full = [], None
    for _ in range(5):
        dist.barrier()
        torch.cuda.synchronize()
        tm = {}
        t0 = time.perf_counter()
        full = par.transform_sharded(fruit, Xd, rank, world, on_device=True, timings=tm)
        torch.cuda.synchronize()
        tm["end_to_end_s"] = time.perf_counter() - t0
        runs.append(tm)
    med = {k: float(np.median([r[k] for r in runs])) for k in ("broadcast_s", "compute_s", "gather_s", "end_to_end_s")}
    slc = fruit.get_slice()
    iss = slc.get_iss()[0]
    strings = [str(w) for w in iss.words]
    depths = [iss._depth(i) for i in range(len(strings))]
    mine_words = par.shard_words(strings, depths, world)[rank]
    pipe = slc._fused(shape[2], indices=mine_words)
    mine = {"rank": rank, "launch_ms": med["compute_s"] * 1e3, "gather_ms": med["gather_s"] * 1e3,
            "broadcast_ms": med["broadcast_s"] * 1e3, "end_to_end_ms": med["end_to_end_s"] * 1e3,
            "K": pipe.plan.rows, "nodes": pipe.plan.nodes,
            "gathered_bytes_per_rank": runs[-1].get("gathered_bytes_per_rank", 0),
            "kernel": "pieces" if pipe.pieces_loaded() else ("own" if pipe.jit_loaded() else "generic")}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank != 0:
        return None
    # rank 0: the unsharded transform of the same batch, every column compared
    cache = fr.cache.SharedSeedCache(None)
    cache.adopt_device_input(Xd)
    slc._attach(cache)
    whole = slc._fused(shape[2])
    chain = slc._fusable_preparation(shape[2])
    assert whole.set_preparation(shape[1], *chain)
    whole.prepare(shape[0])
    lk = iss.lookup_device(Xd)
    ref = torch.empty((shape[0], whole.n_features), dtype=torch.float64, device="cuda")
    fn1 = lambda: whole.run(Xd, lk, feats=ref)
    t_one = _event_time_us(torch, fn1, reps=3) / 1e3
    torch.cuda.synchronize()
    cols_ok = int(((full - ref).abs().max(dim=0).values == 0).sum().item())
    launches = [e["launch_ms"] for e in everyone]
    e2e = max(e["end_to_end_ms"] for e in everyone)
    K_total = whole.plan.rows
    return {
        "workload": "BASELINE configs[3]: of_weight(6,2) EXTENDED + Indices, (8192,3,1024) f64, INC -> ISS -> "
                    "NPI(q=(0.5,1)), END through fruits_amd.parallel.transform_sharded (batch on rank 0 only)",
        "backend": backend, "world_size": dist.get_world_size(), "K": K_total, "features": int(full.shape[1]),
        "rank_launch_ms": [round(v, 3) for v in launches], "slowest_launch_ms": round(max(launches), 3),
        "balance": round(float(np.mean(launches) / max(launches)), 3),
        "nodes_per_rank": [e["nodes"] for e in everyone], "kernels": sorted({e["kernel"] for e in everyone}),
        "broadcast_ms": round(max(e["broadcast_ms"] for e in everyone), 3),
        "allgather_and_column_gather_ms": round(max(e["gather_ms"] for e in everyone), 3),
        "gathered_bytes_per_rank": everyone[0]["gathered_bytes_per_rank"],
        "end_to_end_ms": round(e2e, 3), "elements_per_s": shape[0] * K_total * shape[2] / (e2e * 1e-3),
        "single_rank_unsharded_launch_ms": round(t_one, 3),
        "equals_unsharded_transform": bool(cols_ok == full.shape[1]), "columns_bit_identical": cols_ok,
        "fit_on_root_ms": round(fit_ms, 1),
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
        sec, sweep_rows, det = extras(torch, fr, nat, dev, quick=args.quick_extras)
        # (inside the roofline object: the driver's record keeps that verbatim)
        res["roofline"]["secondary"] = sec
        res["roofline"]["sweep_N_T_static_interp"] = sweep_rows
        res["extras"] = det
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
