"""Diagnostic (timing build): per-segment s_memtime shares of the walk kernel."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["FRUITS_HIP_DEBUG"] = (f"stamps={16 | int(os.environ.get('DBG_EXTRA', '0'))},dbg_bytes={1 << 22},"
                                  f"persist={os.environ.get('DBG_PERSIST', '1')}")
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
N, D, T = [int(v) for v in os.environ.get("TUNE_SHAPE", "2048,3,1024").split(",")]
words = fr.words.of_weight(2, dim=D)
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
X = np.random.default_rng(0).standard_normal((N, D, T))
Xd = nat.to_device(X)
out = torch.empty((plan.rows, N, T), dtype=torch.float64, device=Xd.device)
work = torch.zeros((1 << 22) + 4096, dtype=torch.uint8, device=Xd.device)
for _ in range(3):
    plan.run(Xd, None, out=out, work=work)
torch.cuda.synchronize()
work.zero_()
plan.run(Xd, None, out=out, work=work); torch.cuda.synchronize()
raw = work[:].cpu().numpy().view(np.uint64)
nwaves = min(N, 1536) * 4 if os.environ.get("DBG_PERSIST", "1") != "0" else N * 4
st = raw[: nwaves * 12].reshape(nwaves, 12).astype(np.float64)
names = ["interp", "factors", "scan-local", "lds+barrier", "prefix+final", "stores", "staging", "-"]
tot = st[:, 8]
ok = tot > 0
print(f"waves reporting {int(ok.sum())} of {nwaves}; lifetime cycles: median {np.median(tot[ok]):.0f}  "
      f"min {tot[ok].min():.0f}  max {tot[ok].max():.0f}")
for i, nm in enumerate(names[:7]):
    print(f"  {nm:14s} median {np.median(st[ok, i]):9.0f}  share {np.median(st[ok, i]) / np.median(tot[ok]) * 100:5.1f}%")
rb = raw[: nwaves * 12].reshape(nwaves, 12)[:, 10].astype(np.int64)[ok]
re = raw[: nwaves * 12].reshape(nwaves, 12)[:, 11].astype(np.int64)[ok]
g0 = rb.min()
start, end = (rb - g0) / 100.0, (re - g0) / 100.0        # microseconds (100 MHz)
span = end.max()
print(f"kernel span {span:.1f} us; wave start: median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
print("wave END deciles (us):", [round(float(np.percentile(end, q)), 1) for q in range(10, 101, 10)])
print("wave lifetime deciles (us):", [round(float(np.percentile(end - start, q)), 1) for q in range(10, 101, 10)])
edges = np.linspace(0, span, 21)
alive = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5% slice:", alive)
