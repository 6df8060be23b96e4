"""What the box sustains beyond the 256 MiB Infinity Cache: fills / copies of growing buffers,
and the config-2 walk kernel over growing batches in both output layouts."""
import sys, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
from bench_pipeline import graph_time

def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(ts))

for mb in (151, 302, 604, 1208, 2416):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    tf = ev(lambda: a.fill_(1.0)); tc = ev(lambda: b.copy_(a))
    print(json.dumps({"buffer_MB": mb, "fill_us": round(tf, 1), "fill_TBs": round(n * 8 / tf / 1e6, 3),
                      "copy_us": round(tc, 1), "copy_rw_TBs": round(2 * n * 8 / tc / 1e6, 3)}), flush=True)
    del a, b
words = fr.words.of_weight(2, dim=3)
plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
K = plan.rows
for N in (1024, 1536, 2048, 3072, 4096, 8192, 16384):
    Xd = torch.randn((N, 3, 1024), dtype=torch.float64, device="cuda")
    res = {"N": N, "out_MB": round(K * N * 1024 * 8 / 1e6)}
    for layout in ("KNT", "NKT"):
        out = torch.empty((K, N, 1024) if layout == "KNT" else (N, K, 1024), dtype=torch.float64, device="cuda")
        t = ev(lambda: plan.run(Xd, None, out=out, layout=layout))
        b = 8.0 * N * 1024 * (3 + K)
        res[layout + "_us"] = round(t, 1); res[layout + "_TBs"] = round(b / t / 1e6, 3)
        del out
    print(json.dumps(res), flush=True)
