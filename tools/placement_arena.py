"""Where inside ONE allocation does the 48-word launch (805 MB of output) run fast?  An arena of 8 GiB,
the output placed at offsets of 256 MiB steps (and a few odd ones), every placement timed in three
interleaved rounds; then the same with a second arena.  If the placements of one arena differ among
themselves, what decides is finer than the allocation; if an arena is all of one kind, it is a property
of the allocation call."""
import sys
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
import bench
N, T, K = 2048, 1024, 48
Xd = bench._device_batch(torch, (N, 3, T), 0)
w15 = fr.words.of_weight(2, dim=3)
plan = fr.ISS([w15[i % 15] for i in range(48)])._plan(0, 48)
plan.prepare(N, T)
nbytes = K * N * T * 8
MiB = 1 << 20
def t_us(buf):
    return bench._event_time_us(torch, lambda: plan.run(Xd, None, out=buf), reps=10)
for arena_no in range(2):
    arena = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
    base = arena.data_ptr()
    offs = [i * 256 * MiB for i in range(0, 28, 3)] + [2 * MiB, 34 * MiB, 1000 * MiB + 4096]
    offs = [o for o in offs if o + nbytes <= arena.numel()]
    views = [arena[o:o + nbytes].view(torch.float64).view(K, N, T) for o in offs]
    print(f"arena {arena_no} at {base:#x} (mod 1 GiB {base % (1 << 30):#x})")
    res = [[] for _ in offs]
    for rnd in range(3):
        for i, v in enumerate(views):
            res[i].append(round(t_us(v), 1))
    for o, r in zip(offs, res):
        print(f"  offset {o / MiB:8.1f} MiB: {r}", flush=True)
    keep = arena if arena_no == 0 else None   # (the second arena lies elsewhere)
own = [torch.empty((K, N, T), dtype=torch.float64, device="cuda") for _ in range(4)]
print("four own allocations:", [(hex(b.data_ptr()), round(t_us(b), 1)) for b in own])
