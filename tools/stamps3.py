"""Diagnostic (timing build): unit start / end times of the static-program walk kernel."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["FRUITS_HIP_DEBUG"] = "16"
os.environ["FRUITS_HIP_DBG_BYTES"] = str(1 << 22)
os.environ["FRUITS_HIP_GROUPS"] = "1"
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
N, D, T = int(os.environ.get("STAMP_N", "2048")), 3, 1024
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
rec = raw[: R * 4 * 12].reshape(R, 4, 12).astype(np.int64)
ok = rec[:, 0, 8] == 1
print("workgroups reporting:", int(ok.sum()))
g0 = rec[ok][:, :, 10].min()
us = lambda v: (v - g0) / 100.0
beg, end = us(rec[:, :, 10]), us(rec[:, :, 11])
u0, u1 = us(rec[:, :, 0]), us(rec[:, :, 1])
two = rec[:, 0, 1] > 0
print(f"kernel span {end[ok].max():.1f} us; start median {np.median(beg[ok]):.1f} max {beg[ok].max():.1f}")
for label, m in (("two units", ok & two), ("one unit", ok & ~two)):
    if m.sum() == 0:
        continue
    print(f"{label}: {int(m.sum())} workgroups; end deciles", [round(float(np.percentile(end[m], q)), 1) for q in range(0, 101, 10)])
m = ok & two
if m.sum():
    print("second unit start deciles", [round(float(np.percentile(u1[m], q)), 1) for q in range(0, 101, 10)])
    print("second unit duration deciles", [round(float(np.percentile((end - u1)[m], q)), 1) for q in range(0, 101, 10)])
    print("first unit duration deciles (two-unit wgs)", [round(float(np.percentile((u1 - u0)[m], q)), 1) for q in range(0, 101, 10)])
m = ok & ~two
if m.sum():
    print("first unit duration deciles (one-unit wgs)", [round(float(np.percentile((end - u0)[m], q)), 1) for q in range(0, 101, 10)])
print("by blockIdx range: start median/max, first-unit duration median, end median/max (us)")
for lo in range(0, R, 128):
    sl = slice(lo, lo + 128)
    e1 = np.where(rec[sl, :, 1] > 0, u1[sl], end[sl])
    print(f"  [{lo:4d},{lo + 128:4d})  start {np.median(beg[sl]):5.1f} / {beg[sl].max():5.1f}   unit0 start {np.median(u0[sl]):5.1f}  "
          f"first unit {np.median(e1 - u0[sl]):5.1f}   end {np.median(end[sl]):5.1f} / {end[sl].max():5.1f}")
late = beg[:, 0] > 10
print("workgroups starting later than 10 us:", int(late.sum()), "blockIdx of the first few:", np.nonzero(late)[0][:20].tolist())
hw = rec[:, 0, 4]
xcc = rec[:, 0, 5] & 0xf
# gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (se bits may be wider)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
uniq = np.unique(key)
print("distinct (xcc, se, sh, cu):", len(uniq), " xcc ids:", np.unique(xcc).tolist(), " blockIdx%8 -> xcc:",
      [int(np.bincount(xcc[np.arange(R) % 8 == j]).argmax()) for j in range(8)])
cnt = np.array([(key == k).sum() for k in uniq])
print("workgroups per CU: histogram", np.bincount(cnt).tolist())
k0 = uniq[np.argsort(-cnt)[:3]]
for k in list(k0) + list(uniq[np.argsort(cnt)[:2]]):
    m = np.nonzero(key == k)[0]
    print(f"  cu {k}: blocks {m.tolist()} starts {[round(float(beg[b, 0]), 1) for b in m]} ends {[round(float(end[b, 0]), 1) for b in m]}")
