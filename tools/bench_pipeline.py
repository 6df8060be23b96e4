"""Config 3 / 5 style pipelines: fused launch vs materialising path, device-resident timing."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, '.')
import torch
import fruits_amd as fr
from fruits_amd import _native as nat

def graph_time(fn, reps=10, rounds=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    ts = []
    for _ in range(rounds):
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(ts))

def run(name, N, D, T, words, weighting, sieves):
    X = np.random.default_rng(0).standard_normal((N, D, T))
    fruit = fr.Fruit(name)
    fruit.add(fr.preparation.INC)
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=weighting)
    fruit.add(iss)
    for s in sieves: fruit.add(s)
    for slc in fruit: slc.fit_sample_size = 1.0
    t0 = time.perf_counter(); np.random.seed(0); fruit.fit(X[:256]); t_fit = time.perf_counter() - t0
    t0 = time.perf_counter(); f1 = fruit.transform(X); t_e2e_first = time.perf_counter() - t0
    t0 = time.perf_counter(); f1 = fruit.transform(X); t_e2e = time.perf_counter() - t0
    slc = fruit.get_slice(0)
    cache = fr.cache.SharedSeedCache(X)
    Xd = cache.input_device(X)
    Pd = slc._prepare_device(Xd, cache)
    slc._attach(cache)
    pipe = slc._fused(T)
    lk = iss.lookup_device(Pd)
    feats = torch.empty((N, pipe.n_features), dtype=torch.float64, device=Pd.device)
    wb = int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, N, 0 if lk is None else lk.shape[0]))
    work = torch.empty(max(wb, 1), dtype=torch.uint8, device=Pd.device)
    t_fused = graph_time(lambda: pipe.run(Pd, lk, feats=feats, work=work))
    plan = iss._plan(0, len(words)); K = plan.rows
    if K * N * T * 8 < 40e9:
        out = torch.empty((K, N, T), dtype=torch.float64, device=Pd.device)
        wb2 = plan.workspace_bytes(N, T, 0 if lk is None else lk.shape[0])
        work2 = torch.empty(max(wb2, 1), dtype=torch.uint8, device=Pd.device)
        t_mat = graph_time(lambda: plan.run(Pd, lk, out=out, work=work2))
        del out
    else:
        t_mat = float("nan")
    if K * N * T * 8 < 40e9:
        os.environ["FRUITS_AMD_FUSED"] = "0"; slc._fused_cache = {}
        t0 = time.perf_counter(); f0 = fruit.transform(X); t_e2e_plain = time.perf_counter() - t0
        os.environ["FRUITS_AMD_FUSED"] = "1"; slc._fused_cache = {}
    else:
        f0, t_e2e_plain = f1, float("nan")
    diff = np.abs(f0 - f1)
    eq_bytes = 8.0 * N * T * (plan.dims_used + K)
    res = {"name": name, "N": N, "D": D, "T": T, "W": len(words), "K": K, "F": int(pipe.n_features),
           "fit_s(256 series)": round(t_fit, 3),
           "fused_launch_us": round(t_fused, 1), "materialise_launch_us": round(t_mat, 1),
           "fused_elements_per_s": N * K * T / (t_fused * 1e-6),
           "fused_equiv_materialised_GBs": eq_bytes / (t_fused * 1e-6) / 1e9,
           "materialise_GBs": eq_bytes / (t_mat * 1e-6) / 1e9,
           "transform_e2e_fused_ms": round(t_e2e * 1e3, 2), "transform_e2e_first_ms": round(t_e2e_first * 1e3, 2),
           "transform_e2e_unfused_ms": round(t_e2e_plain * 1e3, 2),
           "max_feature_diff_fused_vs_unfused": float(diff.max()), "frac_diff": float((diff > 0).mean())}
    print(json.dumps(res))

which = (sys.argv[1] if len(sys.argv) > 1 else "cfg3") if __name__ == "__main__" else "none"
if which in ("cfg3", "all"):
    run("cfg3_indices", 2048, 3, 1024, fr.words.of_weight(4, dim=2), fr.iss.weighting.Indices(),
        [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END()])
    run("cfg3_unweighted", 2048, 3, 1024, fr.words.of_weight(4, dim=2), None,
        [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END()])
if which in ("cfg2", "all"):
    run("cfg2_pipeline", 2048, 3, 1024, fr.words.of_weight(2, dim=3), None,
        [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END()])
if which in ("cfg5", "all"):
    run("cfg5_l1", 1024, 6, 4096, fr.words.of_weight(9, dim=1), fr.iss.weighting.L1(),
        [fr.sieving.NPI(), fr.sieving.END()])

def run_arctic(N, D, T, L):
    X = np.random.default_rng(0).standard_normal((N, D, T))
    words = fr.words.alternate_sign([fr.words.SimpleWord(L * "[1]"), fr.words.SimpleWord(L * "[2]"),
                                     fr.words.SimpleWord((L // 2) * "[1][2]"),
                                     fr.words.SimpleWord((L // 2) * "[2][1]")])
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, semiring=fr.semiring.Arctic())
    plan = iss._plan(0, len(words)); K = plan.rows
    Xd = nat.to_device(X)
    out = torch.empty((K, N, T), dtype=torch.float64, device=Xd.device)
    t = graph_time(lambda: plan.run(Xd, None, out=out))
    b = 8.0 * N * T * (plan.dims_used + K)
    print(json.dumps({"name": f"arctic_alt{L}", "N": N, "T": T, "W": len(words), "K": K, "nodes": plan.nodes,
                      "materialise_launch_us": round(t, 1), "GBs": b / (t * 1e-6) / 1e9,
                      "elements_per_s": N * K * T / (t * 1e-6)}))

if which in ("arctic", "all"):
    run_arctic(2048, 3, 1024, 24)

if which in ("cfg4",):
    run("cfg4_indices_1gpu", 8192, 3, 1024, fr.words.of_weight(6, dim=2), fr.iss.weighting.Indices(),
        [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END()])

def run_reduced1(N, D, T):
    """fruit_reduced.py slice 1 verbatim: NEW(INC) -> STD -> ISS(of_weight(4,2), EXT, Indices) ->
    NPI x3 (inc 0,1,2), MPI x3, END; fit_sample_size = 1.0"""
    X = np.random.default_rng(0).standard_normal((N, D, T))
    fruit = fr.Fruit("reduced slice 1")
    fruit.add(fr.preparation.NEW(fr.preparation.INC()))
    fruit.add(fr.preparation.STD)
    fruit.add(fr.ISS(fr.words.of_weight(4, 2), mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices()))
    for inc in (0, 1, 2): fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=inc))
    for inc in (0, 1, 2): fruit.add(fr.sieving.MPI(q=(0.5, 1.0), inc=inc))
    fruit.add(fr.sieving.END)
    for slc in fruit: slc.fit_sample_size = 1.0
    np.random.seed(0)
    t0 = time.perf_counter(); fruit.fit(X); t_fit = time.perf_counter() - t0
    fruit.transform(X)
    t0 = time.perf_counter(); f1 = fruit.transform(X); t_tr = time.perf_counter() - t0
    os.environ["FRUITS_AMD_FUSED"] = "0"; os.environ["FRUITS_AMD_DEVICE_FIT"] = "0"
    for slc in fruit: slc._fused_cache = {}
    t0 = time.perf_counter(); f0 = fruit.transform(X); t_tr0 = time.perf_counter() - t0
    np.random.seed(0)
    t0 = time.perf_counter(); fruit.fit(X); t_fit0 = time.perf_counter() - t0
    os.environ["FRUITS_AMD_FUSED"] = "1"; os.environ["FRUITS_AMD_DEVICE_FIT"] = "1"
    d = np.abs(f0 - f1)
    print(json.dumps({"name": "fruit_reduced_slice1", "N": N, "D": D, "T": T, "features": int(f1.shape[1]),
                      "fit_device_s": round(t_fit, 3), "fit_host_quantiles_s": round(t_fit0, 3),
                      "transform_fused_ms": round(t_tr * 1e3, 2), "transform_unfused_ms": round(t_tr0 * 1e3, 2),
                      "frac_entries_differing": float((d > 1e-9 * (1 + np.abs(f0))).mean())}))

if which in ("reduced1", "all"):
    run_reduced1(2048, 1, 1024)


def build_reduced():
    """experiments/fruit_reduced.py, all four slices."""
    fruit = fr.Fruit("Reduced Fruit")
    iss_r = fr.ISS(fr.words.of_weight(4, 2), mode=fr.ISSMode.EXTENDED,
                   weighting=fr.iss.weighting.Indices())
    iss_a = fr.ISS(fr.words.alternate_sign([
        fr.words.SimpleWord(24 * "[1]"), fr.words.SimpleWord(24 * "[2]"),
        fr.words.SimpleWord(12 * "[1][2]"), fr.words.SimpleWord(12 * "[2][1]")]),
        mode=fr.ISSMode.EXTENDED, semiring=fr.semiring.Arctic())
    cos_words = (list(fr.words.of_weight(1, 2)) + list(fr.words.of_weight(2, 2))
                 + list(fr.words.of_weight(3, 2)))

    def sieves():
        for inc in (0, 1, 2): fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=inc))
        for inc in (0, 1, 2): fruit.add(fr.sieving.MPI(q=(0.5, 1.0), inc=inc))
        fruit.add(fr.sieving.END)
    fruit.cut(); fruit.add(fr.preparation.NEW(fr.preparation.INC())); fruit.add(fr.preparation.STD)
    fruit.add(iss_r); sieves()
    fruit.cut(); fruit.add(fr.preparation.NEW(fr.preparation.INC())); fruit.add(iss_a); sieves()
    for e in (1, 2):
        fruit.cut(); fruit.add(fr.preparation.NEW(fr.preparation.INC())); fruit.add(fr.preparation.STD)
        fruit.add(fr.CosWISS(freqs=[i / 20 for i in range(1, 11, 2)], words=cos_words, exponent=e,
                             total_weighting=True))
        sieves()
    for slc in fruit: slc.fit_sample_size = 1.0
    return fruit


def run_reduced_full(N, T):
    X = np.random.default_rng(0).standard_normal((N, 1, T)).cumsum(axis=2)
    fruit = build_reduced()
    np.random.seed(0)
    t0 = time.perf_counter(); fruit.fit(X); torch.cuda.synchronize(); t_fit = time.perf_counter() - t0
    fruit.transform(X)
    t0 = time.perf_counter(); f1 = fruit.transform(X); t_tr = time.perf_counter() - t0
    per_slice = []
    for i, slc in enumerate(fruit):
        cache = fr.cache.SharedSeedCache(X)
        slc.transform(X, cache=cache)
        t0 = time.perf_counter(); slc.transform(X, cache=cache); per_slice.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(json.dumps({"name": "fruit_reduced_full", "N": N, "T": T, "features": int(f1.shape[1]),
                      "fit_s": round(t_fit, 3), "transform_ms": round(t_tr * 1e3, 2),
                      "per_slice_transform_ms": per_slice,
                      "fused": [s._fused(T) is not None for s in fruit],
                      "nan_features": int(np.isnan(f1).sum())}))

if which in ("reduced", "all"):
    run_reduced_full(2048, 1024)
