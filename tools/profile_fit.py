"""Where Fruit.fit spends its time on experiments/fruit_reduced.py at (2048,1,1024)."""
import cProfile, pstats, sys, time, io
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
import bench_pipeline as bp
X = np.random.default_rng(0).standard_normal((2048, 1, 1024)).cumsum(axis=2)
fruit = bp.build_reduced()
np.random.seed(0); fruit.fit(X); torch.cuda.synchronize()
for rep in range(2):
    np.random.seed(0); t0 = time.perf_counter(); fruit.fit(X); torch.cuda.synchronize(); print("fit s", time.perf_counter() - t0)
fruit.transform(X); t0 = time.perf_counter(); fruit.transform(X); print("transform s", time.perf_counter() - t0)
pr = cProfile.Profile(); np.random.seed(0); pr.enable(); fruit.fit(X); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
