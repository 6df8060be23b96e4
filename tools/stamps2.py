"""Diagnostic (timing build): workgroups that walk two units against those that walk one
(N = 2048 on 1536 resident workgroups): per-segment cycles and end times."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["FRUITS_HIP_DEBUG"] = "16"
os.environ["FRUITS_HIP_DBG_BYTES"] = str(1 << 22)
os.environ["FRUITS_HIP_GROUPS"] = "1"
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
N, D, T = 2048, 3, 1024
words = fr.words.of_weight(2, dim=D)
plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
Xd = nat.to_device(np.random.default_rng(0).standard_normal((N, D, T)))
out = torch.empty((plan.rows, N, T), dtype=torch.float64, device=Xd.device)
work = torch.zeros((1 << 22) + 4096, dtype=torch.uint8, device=Xd.device)
for _ in range(3):
    plan.run(Xd, None, out=out, work=work)
torch.cuda.synchronize()
work.zero_()
plan.run(Xd, None, out=out, work=work); torch.cuda.synchronize()
raw = work[:].cpu().numpy().view(np.uint64)
R = 1536
rec = raw[: R * 4 * 12].reshape(R, 4, 12)
st = rec.astype(np.float64)
names = ["interp", "factors", "scan-local", "lds+barrier", "prefix+final", "stores", "staging"]
rb_all = rec[:, :, 10].astype(np.int64)
print("workgroups not reporting:", int((rb_all == 0).any(axis=1).sum()))
g0 = rb_all[rb_all > 0].min()
start = (rec[:, :, 10].astype(np.int64) - g0) / 100.0
end = (rec[:, :, 11].astype(np.int64) - g0) / 100.0
two = np.arange(R) < (N - R)
for label, m in (("two units", two), ("one unit", ~two)):
    print(f"{label}: {int(m.sum())} workgroups; end median {np.median(end[m]):.1f} us  p10 {np.percentile(end[m], 10):.1f}  "
          f"p90 {np.percentile(end[m], 90):.1f}  max {end[m].max():.1f}; lifetime cycles median {np.median(st[m][:, :, 8]):.0f}")
    for i, nm in enumerate(names):
        print(f"    {nm:14s} median {np.median(st[m][:, :, i]):9.0f} cycles")
print("second unit alone (difference of medians, cycles):",
      {nm: int(np.median(st[two][:, :, i]) - np.median(st[~two][:, :, i])) for i, nm in enumerate(names)})
print("by blockIdx range: start median / end median / end max (us), lifetime cycles median")
for lo in range(0, R, 128):
    sl = slice(lo, lo + 128)
    print(f"  [{lo:4d},{lo + 128:4d})  start {np.median(start[sl]):6.1f}  end {np.median(end[sl]):6.1f}  max {end[sl].max():6.1f}  "
          f"cycles {np.median(st[sl][:, :, 8]):8.0f}  staging {np.median(st[sl][:, :, 6]):7.0f}  stores {np.median(st[sl][:, :, 5]):7.0f}")
print("clock: lifetime cycles / lifetime us =", float(np.median(st[:, :, 8] / ((end - start) * 1.0))), "cycles per us")
