"""`python tools/run_mat.py`: the materialising walk through the record interpreter (no static
programs) on a few plans: of_weight(2,3) at the headline shape and at N = 8192, of_weight(4,2)
(K = 115) unweighted and with Indices weighting."""
import os, sys
os.environ["FRUITS_HIP_STATIC"] = "0"
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench
for name, words, w, N in (("w2", fr.words.of_weight(2, dim=3), None, 2048), ("w2", fr.words.of_weight(2, dim=3), None, 8192),
                          ("w4", fr.words.of_weight(4, dim=2), None, 2048),
                          ("w4+Indices", fr.words.of_weight(4, dim=2), fr.iss.weighting.Indices(), 2048)):
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=w)
    plan = iss._plan(0, len(words))
    Xd = bench._device_batch(torch, (N, 3, 1024), 0)
    lk = None if w is None else iss.lookup_device(Xd)
    out = torch.empty((plan.rows, N, 1024), dtype=torch.float64, device="cuda")
    t = bench._event_time_us(torch, lambda: plan.run(Xd, lk, out=out), reps=20)
    b = 8.0 * N * 1024 * (plan.dims_used + plan.rows)
    print(f"{name} N={N} K={plan.rows}: {t:.1f} us  {b / t / 1e3:.0f} GB/s = {b / t / 8e6:.3f} of 8 TB/s")
