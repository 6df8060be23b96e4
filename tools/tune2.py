"""A/B timing of iss_walk variants (env knobs) in ONE process, interleaved rounds.
Launches are captured into a HIP graph (20 per replay) so host launch overhead
does not hide kernel time."""
import os, sys, json
import numpy as np
sys.path.insert(0, '.')
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
N, D, T = [int(v) for v in os.environ.get("TUNE_SHAPE", "2048,3,1024").split(",")]
wspec = os.environ.get("TUNE_WORDS", "2,3")
ww, wd = [int(v) for v in wspec.split(",")]
words = fr.words.of_weight(ww, dim=wd)
if os.environ.get("TUNE_TILE48"):       # the metric's "48 weight-2 words": words[i % 15], SINGLE mode
    words = [words[i % len(words)] for i in range(48)]
    iss = fr.ISS(words)
else:
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
K = plan.rows
X = np.random.default_rng(0).standard_normal((N, D, T))
Xd = nat.to_device(X)
out = torch.empty((K, N, T), dtype=torch.float64, device=Xd.device)
variants = json.loads(sys.argv[1])
REPS = 20
graphs = []
for v in variants:
    for k, val in v.items():
        os.environ[k] = str(val)
    plan.run(Xd, None, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                plan.run(Xd, None, out=out)
    graphs.append(g)
res = {i: [] for i in range(len(variants))}
for rnd in range(8):
    for i, g in enumerate(graphs):
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) / REPS * 1e3)
balg = 8.0 * N * T * (plan.dims_used + K)
for i, v in enumerate(variants):
    med = float(np.median(res[i])); mn = float(np.min(res[i]))
    print(f"{v}  median {med:7.1f} us  min {mn:7.1f} us  -> {balg/med/1e3:7.1f} GB/s ({balg/med/1e3/8000:.3f})")
