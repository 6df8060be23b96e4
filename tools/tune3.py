import os, sys, json
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
def bench(pad_k, pad_n, reps=20):
    sn = T + pad_n
    sk = N * sn + pad_k
    out = torch.empty(K * sk + 16, dtype=torch.float64, device=Xd.device)
    ts = []
    for _ in range(6):
        plan.run(Xd, None, out=out, strides=(sk, sn)); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            plan.run(Xd, None, out=out, strides=(sk, sn))
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(ts))
for pad_k, pad_n in [(0, 0), (32, 0), (512, 0), (1024+32, 0), (8192+64, 0), (0, 2), (0, 16), (0, 32), (0, 64), (32, 32)]:
    print(f"pad_k={pad_k:6d} pad_n={pad_n:4d}: {bench(pad_k, pad_n):7.1f} us")
