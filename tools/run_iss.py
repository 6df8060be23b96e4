"""Runs the config-2 ISS launch a few times (profiling target)."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
N, D, T = 2048, 3, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
words = fr.words.of_weight(2, dim=D)
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
X = np.random.default_rng(0).standard_normal((N, D, T))
Xd = nat.to_device(X)
out = torch.empty((plan.rows, N, T), dtype=torch.float64, device=Xd.device)
for _ in range(reps):
    plan.run(Xd, None, out=out)
torch.cuda.synchronize()
