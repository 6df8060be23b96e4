"""`python tools/select_bench.py [rows] [reps]`: fr_select_ranks alone on a (rows, 2048, 1024) block that
looks like a fit's (iterated sums of a random walk's increments), with fruit_reduced's jobs - the
median (+ its neighbour) and the maximum of the values and of their first and second differences -
ms per call (wall, synchronised) and the effective read rate of ONE pass over the block.
FRUITS_HIP_LIB selects the build (tools/gpu_ab.sh style A/B)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from fruits_amd import _native as nat
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, T = 2048, 1024
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((N, T), device="cuda", dtype=torch.float64, generator=g)
block = torch.empty((rows, N, T), device="cuda", dtype=torch.float64)
cur = x.clone()
for r in range(rows):                       # rows of growing "depth", like a trie's
    cur = torch.cumsum(cur * x, dim=1) if r % 3 else torch.cumsum(x * (r + 1), dim=1)
    block[r] = cur / (cur.abs().max() + 1e-300) * 10.0 ** (r % 7)
n = N * T
jr, ji, jk = [], [], []
for r in range(rows):
    for inc in (0, 1, 2):
        for k in ((n - 1) // 2, (n - 1) // 2 + 1, n - 1):
            jr.append(r); ji.append(inc); jk.append(k)
ref = None
ts = []
for i in range(reps + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = nat.select_ranks(block, jr, ji, jk)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts = ts[2:]
# spot check against torch on a few rows
for r in (0, rows // 2, rows - 1):
    v = block[r].flatten().sort().values
    want = [float(v[(n - 1) // 2]), float(v[(n - 1) // 2 + 1]), float(v[n - 1])]
    got = [float(out[jr.index(r) + j]) for j in range(3)]
    assert want == got, (r, want, got)
ms = float(np.median(ts)) * 1e3
print(f"rows {rows} jobs {len(jr)}: {ms:.2f} ms per call (min {min(ts) * 1e3:.2f}), "
      f"{rows * n * 8 / 1e9 / (ms / 1e3) / 1e3:.2f} TB/s if it were one pass over the block")
