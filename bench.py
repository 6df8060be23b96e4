#!/usr/bin/env python3
"""Benchmark of the ISS hot path on MI355X (the metric of BASELINE.json).

One "step" = one pass of the hot path over one synthetic batch: the iterated
sums of ``of_weight(2, dim=3)`` in EXTENDED mode (W = 15 words, K = 18 sums) of a
``(2048, 3, 1024)`` float64 batch that is already resident in HBM, written as the
reference's ``(K, N, T)`` tensor (BASELINE.json configs[1]).  Prints ONE JSON
line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU;
every rank processes its own batch (the path is independent per series, so it
shards with no data-path collective: weak scaling), the timed region is
bracketed by barrier + synchronize and the slowest rank's time counts.
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


def cpu_baseline(words, seconds_budget: float = 15.0):
    """The C oracle (oracle/iss_oracle.c, the reference algorithm restated) on
    the host cores, on a bounded sample of the same workload."""
    from oracle import c_oracle as corc
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
    reps, t_total, K = 0, 0.0, 18
    out = None
    while t_total < seconds_budget and reps < 50:
        t0 = time.perf_counter()
        out = corc.iss_transform(X, strs, "EXTENDED", out=out, nthreads=threads)
        t_total += time.perf_counter() - t0
        reps += 1
    K = out.shape[0]
    value = n_sample * K * N_STEPS_T * reps / t_total
    return {
        "value": value, "unit": "iterated-sum elements/s", "cores": threads, "kind": "port",
        "sample": f"{n_sample} of {N_SERIES} series x {reps} reps, oracle/iss_oracle.c "
                  f"(OpenMP over series, {threads} threads)",
    }


def _event_time_us(torch, fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def extras(torch, fr, nat, dev):
    """Secondary measurements on the same GPU (not the headline value): the metric's
    "48 words" reading, and the fused INC->ISS->NPI,END pipeline of BASELINE configs[2]."""
    out = {}
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N_SERIES, N_DIMS, N_STEPS_T))
    Xd = nat.to_device(X)
    # (a) words[i % 15] tiled to 48, SINGLE mode: K = 48 rows, 855.6 MB algorithmic
    w15 = fr.words.of_weight(2, dim=N_DIMS)
    w48 = [w15[i % 15] for i in range(48)]
    plan = fr.ISS(w48)._plan(0, 48)
    buf = torch.empty((48, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)
    t = _event_time_us(torch, lambda: plan.run(Xd, None, out=buf))
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (3 + 48)
    out["words48_single"] = {"kernel_us": t, "elements_per_s": N_SERIES * 48 * N_STEPS_T / (t * 1e-6),
                             "GBs": b_alg / (t * 1e-6) / 1e9, "frac": b_alg / (t * 1e-6) / 1e9 / HBM_PEAK_GBS}
    del buf
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
    # (b) config 3 shape: INC -> ISS(of_weight(4,2) EXTENDED, Indices) -> NPI(q=(.5,1)), END, fused
    fruit = fr.Fruit("cfg3")
    fruit.add(fr.preparation.INC)
    iss = fr.ISS(fr.words.of_weight(4, dim=2), mode=fr.ISSMode.EXTENDED,
                 weighting=fr.iss.weighting.Indices())
    fruit.add(iss)
    fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(0)
    fruit.fit(X[:128])
    slc = fruit.get_slice()
    cache = fr.cache.SharedSeedCache(X)
    cache.adopt_device_input(Xd)
    Pd = slc._prepare_device(Xd, cache)
    slc._attach(cache)
    pipe = slc._fused(N_STEPS_T)
    lk = iss.lookup_device(Pd)
    feats = torch.empty((N_SERIES, pipe.n_features), dtype=torch.float64, device=dev)
    t = _event_time_us(torch, lambda: pipe.run(Pd, lk, feats=feats))
    K = pipe.plan.rows
    fruit.transform(X)   # first call builds plans / uploads tables
    t0 = time.perf_counter()
    fruit.transform(X)
    e2e = time.perf_counter() - t0
    out["config3_fused_pipeline"] = {
        "launch_us": t, "K": K, "features": pipe.n_features,
        "elements_per_s": N_SERIES * K * N_STEPS_T / (t * 1e-6),
        "algorithmic_bytes": 8.0 * N_SERIES * N_STEPS_T * 2 + 8.0 * N_SERIES * pipe.n_features,
        "equivalent_materialised_GBs": 8.0 * N_SERIES * N_STEPS_T * (2 + K) / (t * 1e-6) / 1e9,
        "fruit_transform_end_to_end_ms": e2e * 1e3,
        "note": "no (K,N,T) tensor is written; the GB/s figure is what a materialising run "
                "of the same work would have needed, not achieved bandwidth",
    }
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--groups", type=int, default=0)
    args = ap.parse_args()

    import torch
    import fruits_amd as fr
    from fruits_amd import _native as nat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or bool(os.environ.get("FRUITS_BENCH_FORCE_DIST"))
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = nat.require_device()

    words = fr.words.of_weight(2, dim=N_DIMS)
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
    plan = iss._plan(0, len(words))
    K = plan.rows
    rng = np.random.default_rng(rank)
    X = rng.standard_normal((N_SERIES, N_DIMS, N_STEPS_T))
    Xd = nat.to_device(X)
    out = torch.empty((K, N_SERIES, N_STEPS_T), dtype=torch.float64, device=dev)

    def step():
        plan.run(Xd, None, out=out, layout="KNT", groups=args.groups)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    # Kernel duration: ONE pair of HIP events around the K back-to-back launches of the
    # timed region, on the stream the kernel is launched on (torch's current stream is
    # passed through the C ABI); average = elapsed / K, an upper bound that still contains
    # the launch gaps.  (An event pair per step inserts two barrier packets between
    # consecutive kernels: measured 78 us/step and "72.6 us" per kernel where this loop
    # takes 70.2 us per step and rocprofv3 reports 68.1 us per kernel.)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record()
    for _ in range(args.steps):
        step()
    ev_b.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_avg_s = ev_a.elapsed_time(ev_b) / 1e3 / args.steps

    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the timed kernel really produced the tensor (one row against numpy)
    row0 = out[0, :4].cpu().numpy()
    assert np.allclose(row0, np.cumsum(X[:4, 0] ** 2, axis=1), rtol=1e-9)

    elements = N_SERIES * K * N_STEPS_T
    value = elements * args.steps * world / elapsed
    d_used = plan.dims_used
    b_alg = 8.0 * N_SERIES * N_STEPS_T * (d_used + K)  # read X once, write every sum once
    achieved = b_alg / kernel_avg_s / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                traffic = json.load(f).get("iss_walk_bytes_per_launch")
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
            "sharding": "series (one batch per GPU), no data-path collective",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "iss_walk_kernel", "algorithmic_bytes_per_launch": b_alg,
            "kernel_avg_us": kernel_avg_s * 1e6,
            "timing": "one HIP event pair around the K launches of the timed region / K",
        },
    }
    if rank == 0 and world == 1 and not args.no_extras:
        res["extras"] = extras(torch, fr, nat, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(words)
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
