"""Fused launch of every slice of experiments/fruit_reduced.py at (2048,1,1024)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import fruits_amd as fr
import bench_pipeline as bp
from bench_pipeline import graph_time
N, T = 2048, 1024
X = np.random.default_rng(0).standard_normal((N, 1, T)).cumsum(axis=2)
fruit = bp.build_reduced()
np.random.seed(0); fruit.fit(X[:256])
for i, slc in enumerate(fruit):
    cache = fr.cache.SharedSeedCache(X)
    Pd = slc._prepare_device(cache.input_device(X), cache); slc._attach(cache)
    pipe = slc._fused(T)
    lk = slc.get_iss()[0].lookup_device(Pd)
    feats = torch.empty((N, pipe.n_features), dtype=torch.float64, device="cuda")
    t = graph_time(lambda: pipe.run(Pd, lk, feats=feats), reps=5, rounds=3)
    print(f"slice {i}: K={pipe.plan.rows} F={pipe.n_features} fused launch {t:.0f} us", flush=True)
