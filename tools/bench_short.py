"""Short series: cooperative kernel (one series per workgroup) vs wave-per-series kernel."""
import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import fruits_amd as fr
from bench_pipeline import graph_time
words = fr.words.of_weight(2, 3)
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words)); K = plan.rows
for N, T in ((8192, 256), (8192, 250), (16384, 128), (16384, 100), (32768, 64), (400, 128), (2048, 512)):
    X = torch.randn((N, 3, T), dtype=torch.float64, device="cuda")
    out = torch.empty((K, N, T), dtype=torch.float64, device="cuda")
    t = graph_time(lambda: plan.run(X, None, out=out))
    print(f"PACKED={os.environ.get('FRUITS_HIP_DEBUG', 'packed=1')} N={N} T={T}: {t:.1f} us "
          f"{8.0*N*T*(3+K)/t/1e6:.2f} TB/s", flush=True)
# fused pipeline on short series
X = np.random.default_rng(0).standard_normal((8192, 3, 128))
fruit = fr.Fruit(); fruit.add(fr.preparation.INC)
fruit.add(fr.ISS(fr.words.of_weight(4, 2), mode=fr.ISSMode.EXTENDED))
fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
np.random.seed(0); fruit.fit(X[:256])
slc = fruit.get_slice(0); cache = fr.cache.SharedSeedCache(X)
Pd = slc._prepare_device(cache.input_device(X), cache); pipe = slc._fused(128)
feats = torch.empty((8192, pipe.n_features), dtype=torch.float64, device="cuda")
t = graph_time(lambda: pipe.run(Pd, None, feats=feats))
print(f"PACKED={os.environ.get('FRUITS_HIP_DEBUG', 'packed=1')} fused of_weight(4,2) (8192,3,128): {t:.1f} us", flush=True)
