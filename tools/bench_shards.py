"""Config 4 (fruit_general slice 1, of_weight(6,2) + Indices, (8192,3,1024)) sharded over 1/2/4/8
ranks, every rank's fused launch (raw batch in, INC in the staging) timed ALONE on ONE GPU
(loop-back): word shards (every rank the whole batch and its share of the word list) and series
shards (every rank the whole word list on N / world rows).  The multi-GPU time of the compute part
is the slowest rank's; the all-gather volume per link follows from the block widths."""
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import fruits_amd as fr
from fruits_amd import _native as nat, parallel as par
import bench

p = bench._config4(torch, fr, nat)
N, T = p.N, p.T


def timed(fn, reps=3):
    return bench._event_time_us(torch, fn, reps=reps) / 1e3


fn1, _, pipe1 = p.launch()
t_one = timed(fn1)
print(json.dumps({"unsharded_ms": round(t_one, 2), "K": pipe1.plan.rows, "nodes": pipe1.plan.nodes}), flush=True)
for world in (2, 4, 8):
    parts = par.shard_words(p.strings, p.depths, world)
    times, widths, nodes = [], [], []
    for r in range(world):
        fn, _, pipe = p.launch(indices=parts[r])
        times.append(timed(fn))
        widths.append(pipe.n_features)
        nodes.append(pipe.plan.nodes)
    series = []
    for r in range(world):
        rows = par.shard_series(N, r, world)
        Xr = p.Xd[rows].contiguous()
        pipe = p.slc._fused(T)
        pipe.set_preparation(p.D, *p.chain)
        feats = torch.empty((Xr.shape[0], pipe.n_features), dtype=torch.float64, device="cuda")
        series.append(timed(lambda: pipe.run(Xr, p.lk, feats=feats)))
    gather_mb = max(widths) * N * 8 / 1e6
    print(json.dumps({"world": world, "word_rank_ms": [round(t, 2) for t in times],
                      "word_slowest_ms": round(max(times), 2),
                      "word_slowest_over_ideal": round(max(times) / (t_one / world), 3),
                      "nodes_per_rank": nodes, "nodes_over_ideal": round(max(nodes) / (pipe1.plan.nodes / world), 3),
                      "series_slowest_ms": round(max(series), 2),
                      "series_slowest_over_ideal": round(max(series) / (t_one / world), 3),
                      "block_MB_per_rank": round(gather_mb, 1),
                      "allgather_ms_at_153GBs_per_link": round(gather_mb / 153e3 * 1e3, 3)}), flush=True)
