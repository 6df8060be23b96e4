"""`python tools/fit_profile.py [reps]`: Fruit.fit of experiments/fruit_reduced.py on (2048,1,1024),
wall time per call and a cProfile of the host side (run it under rocprofv3 --kernel-trace --stats
for the kernels' share)."""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
sys.argv, reps = [sys.argv[0], "none"], int(sys.argv[1]) if len(sys.argv) > 1 else 5
import bench_pipeline as bp
X = np.random.default_rng(0).standard_normal((2048, 1, 1024)).cumsum(axis=2)
fruit = bp.build_reduced()
np.random.seed(0); fruit.fit(X); torch.cuda.synchronize()
ts = []
pr = cProfile.Profile()
for _ in range(reps):
    np.random.seed(0)
    t0 = time.perf_counter(); pr.enable(); fruit.fit(X); torch.cuda.synchronize(); pr.disable()
    ts.append(time.perf_counter() - t0)
print("fit ms:", [round(t * 1e3, 1) for t in ts])
for i, slc in enumerate(fruit):
    cache = bp.fr.cache.SharedSeedCache(X)
    np.random.seed(0); slc.fit(X, cache=cache); torch.cuda.synchronize()
    t0 = time.perf_counter(); np.random.seed(0); slc.fit(X, cache=cache); torch.cuda.synchronize()
    print(f"slice {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
