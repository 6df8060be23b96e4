"""`python tools/placement_stride.py`: the distance between the K planes of a materialised (K, N, T) tensor
(the `out_k_stride` of fr_iss_run) against the launch time.  A contiguous tensor's planes lie N T 8 bytes
apart - a power of two for the usual shapes, 16 or 64 MiB - and the K concurrent write streams of a
workgroup then agree in every low address bit.  Every (shape, pad) is timed in two interleaved rounds on one
arena."""
import sys
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench
MiB = 1 << 20
w15 = fr.words.of_weight(2, dim=3)
CASES = [("48 words", fr.ISS([w15[i % 15] for i in range(48)]), 2048, 1024),
         ("of_weight(4,2) K=115", fr.ISS(fr.words.of_weight(4, dim=2), mode=fr.ISSMode.EXTENDED), 2048, 1024),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 2048, 1024),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 8192, 1024),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 2048, 4096),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 8192, 256),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 2048, 256),
         ("of_weight(2,3)", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), 16384, 256)]
PADS = [0, 4096, 65536, MiB // 4, MiB, 2 * MiB, 3 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 8 * MiB + 4096, 12 * MiB, 24 * MiB]
if len(sys.argv) > 1:
    PADS = [int(float(a) * MiB) for a in sys.argv[1:]]
for name, iss, N, T in CASES:
    Xd = bench._device_batch(torch, (N, 3, T), 0)
    plan = iss._plan(0, len(iss.words))
    K = plan.rows
    plan.prepare(N, T)
    arena = torch.empty(K * (N * T + max(PADS) // 8) + 1024, dtype=torch.float64, device="cuda")
    res = {p: [] for p in PADS}
    for rnd in range(2):
        for p in PADS:
            sk = N * T + p // 8
            res[p].append(bench._event_time_us(torch, lambda: plan.run(Xd, None, out=arena, strides=(sk, T)), reps=10))
    b = 8.0 * N * T * (plan.dims_used + K)
    base = min(res[0])
    print(f"{name} ({N},3,{T}) K={K}: planes {N * T * 8 / MiB:.0f} MiB apart, contiguous {base:.1f} us = {b / base / 8e6:.3f}")
    print("   " + "  ".join(f"+{p / MiB:g}M {min(res[p]) / base - 1:+.1%}" for p in PADS[1:]), flush=True)
    del arena, Xd
    torch.cuda.empty_cache()
