"""`python tools/run_cell.py N T [reps]`: one cell of bench.py's sweep - ISS(of_weight(2, 3), EXTENDED)
materialised on a resident (N, 3, T) batch - launched `reps` times (for rocprofv3), us per launch and
the fraction of 8 TB/s."""
import sys
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench
N, T = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
words = fr.words.of_weight(2, dim=3)
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
Xd = bench._device_batch(torch, (N, 3, T), 0)
out = torch.empty((plan.rows, N, T), dtype=torch.float64, device="cuda")
plan.prepare(N, T) if hasattr(plan, "prepare") else None
t = bench._event_time_us(torch, lambda: plan.run(Xd, None, out=out), reps=reps)
b = 8.0 * N * T * (plan.dims_used + plan.rows)
print(f"N={N} T={T} K={plan.rows}: {t:.1f} us = {b / t / 8e6:.3f} of 8 TB/s")
