"""The static program's two launch modes (one group + non-temporal input for cache-sized batches;
three groups and four workgroups per CU beyond) and the walk without a static program, over N at
T = 1024: python tools/static_window.py"""
import os, sys
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench
w2 = fr.words.of_weight(2, dim=3)
plan = fr.ISS(w2, mode=fr.ISSMode.EXTENDED)._plan(0, len(w2))
K = plan.rows
for N in (1536, 2048, 2304, 2560, 2816, 3072, 3584, 4096, 6144):
    Xs = bench._device_batch(torch, (N, 3, 1024), 1)
    buf = torch.empty((K, N, 1024), dtype=torch.float64, device="cuda")
    row = []
    for mode, env in (("static cache<=2.0", {"FRUITS_HIP_DEBUG": "static_cache_x100=200"}),
                      ("static cache<=1.4", {"FRUITS_HIP_DEBUG": "static_cache_x100=140"}),
                      ("static stream", {"FRUITS_HIP_DEBUG": "static_cache_x100=0"}),
                      ("no static", {"FRUITS_HIP_STATIC": "0"})):
        os.environ.pop("FRUITS_HIP_DEBUG", None)
        os.environ.pop("FRUITS_HIP_STATIC", None)
        os.environ.update(env)
        plan.prepare(N, 1024)
        t = bench._event_time_us(torch, lambda: plan.run(Xs, None, out=buf), reps=10)
        row.append(f"{mode} {8.0 * N * 1024 * (3 + K) / (t * 1e-6) / 8e12:.3f}")
    print(N, " | ".join(row), flush=True)
    del Xs, buf
    torch.cuda.empty_cache()
