"""Config 4 (fruit_general slice 1, of_weight(6,2) + Indices, (8192,3,1024)) word-sharded over
1/2/4/8 ranks, every rank's fused launch timed on ONE GPU (loop-back): the 8-GPU time of the
compute part is the slowest rank's; the all-gather volume per link follows from the block widths."""
import sys, json, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import fruits_amd as fr
from fruits_amd import parallel as par
from bench_pipeline import graph_time
N, D, T = 8192, 3, 1024
X = np.random.default_rng(0).standard_normal((N, D, T))
fruit = fr.Fruit("general slice 1")
fruit.add(fr.preparation.INC)
iss = fr.ISS(fr.words.of_weight(6, 2), mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices())
fruit.add(iss)
fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
for s in fruit: s.fit_sample_size = 1.0
np.random.seed(0); fruit.fit(X[:128])
slc = fruit.get_slice(0); cache = fr.cache.SharedSeedCache(X)
Pd = slc._prepare_device(cache.input_device(X), cache); slc._attach(cache)
lk = iss.lookup_device(Pd)
strings = [str(w) for w in iss.words]
depths = [iss._depth(i) for i in range(len(strings))]
per_sum = sum(s.nfeatures() for s in slc.get_sieves())
for world in (1, 2, 4, 8):
    parts = par.shard_words(strings, depths, world)
    times, widths = [], []
    for r in range(world):
        pipe = slc._fused(T, indices=parts[r])
        feats = torch.empty((N, pipe.n_features), dtype=torch.float64, device="cuda")
        times.append(graph_time(lambda: pipe.run(Pd, lk, feats=feats), reps=3, rounds=3) / 1e3)
        widths.append(pipe.n_features)
        del feats
    gather_mb = max(widths) * N * 8 / 1e6
    print(json.dumps({"world": world, "rank_ms": [round(t, 2) for t in times],
                      "slowest_ms": round(max(times), 2), "balance": round(np.mean(times) / max(times), 3),
                      "speedup_vs_1": None, "block_MB_per_rank": round(gather_mb, 1),
                      "allgather_ms_at_153GBs_per_link": round(gather_mb / 153e3 * 1e3, 3)}), flush=True)
