"""Fused vs materialised preparation per slice of experiments/fruit_reduced.py on (2048,1,T), every
pipeline prepared first (own kernels): python tools/prep_modes.py [T]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sys.argv = [sys.argv[0], "none"]
import bench_pipeline as bp
import bench
os.environ["FRUITS_AMD_AUTO_PREPARE"] = "all"
X = np.random.default_rng(0).standard_normal((2048, 1, T)).cumsum(axis=2)
fruit = bp.build_reduced()
np.random.seed(0); fruit.fit(X)
for prep in ("0", "1", "2", "0", "1", "2"):
    os.environ["FRUITS_AMD_FUSED_PREP"] = prep
    row = []
    for i, slc in enumerate(fruit):
        slc._fused_cache = {}
        cache = bp.fr.cache.SharedSeedCache(X); cache.input_device(X)
        slc.transform_device(X, cache=cache); torch.cuda.synchronize()
        row.append(round(bench._event_time_us(torch, lambda: slc.transform_device(X, cache=cache), reps=10)))
    print(f"T={T} FUSED_PREP={prep}: us per slice {row}", flush=True)
