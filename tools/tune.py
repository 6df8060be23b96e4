"""A/B timing of iss_walk variants in ONE process (interleaved rounds)."""
import os, sys, itertools, json
import numpy as np
sys.path.insert(0, '.')
import torch
import fruits_amd as fr
from fruits_amd import _native as nat

N, D, T = 2048, 3, 1024
words = fr.words.of_weight(2, dim=D)
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
K = plan.rows
X = np.random.default_rng(0).standard_normal((N, D, T))
Xd = nat.to_device(X)
out = torch.empty((K, N, T), dtype=torch.float64, device=Xd.device)
variants = []
for g in [1, 2, 3, 4, 9]:
    for nt in [0, 1]:
        variants.append({"FRUITS_HIP_GROUPS": str(g), "FRUITS_HIP_NT": str(nt)})
extra = os.environ.get("TUNE_EXTRA")
if extra:
    variants = [dict(v, **e) for v in variants for e in json.loads(extra)]
res = {i: [] for i in range(len(variants))}
def run(v, reps=20):
    for k, val in v.items():
        os.environ[k] = val
    plan.run(Xd, None, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan.run(Xd, None, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rnd in range(6):
    for i, v in enumerate(variants):
        res[i].append(run(v))
balg = 8.0 * N * T * (D + K)
for i, v in enumerate(variants):
    med = float(np.median(res[i])); mn = float(np.min(res[i]))
    print(f"{v}  median {med:7.1f} us  min {mn:7.1f} us  -> {balg/med/1e3:7.1f} GB/s ({balg/med/1e3/8000:.3f})")
