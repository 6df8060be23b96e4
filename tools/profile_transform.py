"""Where Fruit.transform spends its time on experiments/fruit_reduced.py at (2048,1,1024)."""
import cProfile, pstats, sys, time, io
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
import bench_pipeline as bp
X = np.random.default_rng(0).standard_normal((2048, 1, 1024)).cumsum(axis=2)
fruit = bp.build_reduced()
np.random.seed(0); fruit.fit(X); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); F = fruit.transform(X); print("transform s", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); F = fruit.transform(X); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:4000])
