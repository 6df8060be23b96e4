"""Profiling targets: `python tools/run_kernels.py <which> [reps]` runs `reps` launches of one kernel
on resident data.  which: cfg2 (headline materialising launch), cfg3 / cfg4 / cfg5 (the fused
pipelines of BASELINE configs[2..4] on one GPU), cos1 / cos2 (the CosWISS slices of
experiments/fruit_reduced.py, exponent 1 / 2, materialised)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
import bench

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if which == "cfg2":
    words = fr.words.of_weight(2, dim=3)
    plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
    Xd = bench._device_batch(torch, (2048, 3, 1024), 0)
    out = torch.empty((plan.rows, 2048, 1024), dtype=torch.float64, device="cuda")
    fn = lambda: plan.run(Xd, None, out=out)
elif which in ("cfg3", "cfg4", "cfg5"):
    if which == "cfg3":
        p = bench._Pipeline(torch, fr, nat, (2048, 3, 1024), fr.words.of_weight(4, dim=2),
                            fr.iss.weighting.Indices(), [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END],
                            n_fit=128)
    elif which == "cfg4":
        p = bench._config4(torch, fr, nat)
    else:
        p = bench._Pipeline(torch, fr, nat, (8192, 6, 4096), fr.words.of_weight(9, dim=1),
                            fr.iss.weighting.L1(), [fr.sieving.NPI, fr.sieving.END], n_fit=32)
    fn, _, _ = p.launch()
else:
    e = int(which[-1])
    words = fr.words.of_weight(1, 2) + fr.words.of_weight(2, 2) + fr.words.of_weight(3, 2)
    cw = fr.CosWISS(words, [i / 20 for i in range(1, 11, 2)], exponent=e, total_weighting=True)
    Xd = bench._device_batch(torch, (2048, 2, 1024), 0)
    out = cw.transform_device(Xd)
    fn = lambda: cw.transform_device(Xd, out=out)
fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    fn()
b.record()
torch.cuda.synchronize()
print(f"{which}: {a.elapsed_time(b) / reps * 1e3:.1f} us per launch ({reps} launches)")
