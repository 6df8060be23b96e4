"""`python tools/run_mat.py`: the materialising walk WITHOUT static programs on a few plans, the
record interpreter (FRUITS_HIP_DEBUG=lean=0) and the fused walk's node loop with a store epilogue
(lean=1, the default where it applies) timed alternately in one process."""
import os, sys
os.environ["FRUITS_HIP_STATIC"] = "0"
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench

CASES = (
    ("w2", fr.words.of_weight(2, dim=3), None, None, 2048, 1024),
    ("w2", fr.words.of_weight(2, dim=3), None, None, 8192, 1024),
    ("w4", fr.words.of_weight(4, dim=2), None, None, 2048, 1024),
    ("w4", fr.words.of_weight(4, dim=2), None, None, 8192, 1024),
    ("w4+Indices", fr.words.of_weight(4, dim=2), fr.iss.weighting.Indices(), None, 2048, 1024),
    ("w4+Indices total", fr.words.of_weight(4, dim=2), fr.iss.weighting.Indices(total=True), None, 2048, 1024),
    ("w4 Arctic", fr.words.of_weight(4, dim=2), None, fr.semiring.Arctic(), 2048, 1024),
    ("w4 T=4096", fr.words.of_weight(4, dim=2), None, None, 512, 4096),
    ("w4+Indices T=4096", fr.words.of_weight(4, dim=2), fr.iss.weighting.Indices(), None, 512, 4096),
    ("w3 T=700", fr.words.of_weight(3, dim=3), None, None, 4096, 700),
)
for name, words, w, semi, N, T in CASES:
    kw = {} if semi is None else {"semiring": semi}
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=w, **kw)
    plan = iss._plan(0, len(words))
    Xd = bench._device_batch(torch, (N, 3, T), 0)
    lk = None if w is None else iss.lookup_device(Xd)
    out = torch.empty((plan.rows, N, T), dtype=torch.float64, device="cuda")
    b = 8.0 * N * T * (plan.dims_used + plan.rows)
    best = {}
    for rep in range(2):
        for lean in (1, 0):
            os.environ["FRUITS_HIP_DEBUG"] = f"lean={lean}" + os.environ.get("RUN_MAT_EXTRA", "")
            t = bench._event_time_us(torch, lambda: plan.run(Xd, lk, out=out), reps=10)
            best[lean] = min(best.get(lean, 1e30), t)
    print(f"{name} N={N} T={T} K={plan.rows}: lean {best[1]:.1f} us = {b / best[1] / 8e6:.3f}, "
          f"interpreter {best[0]:.1f} us = {b / best[0] / 8e6:.3f} of 8 TB/s", flush=True)
