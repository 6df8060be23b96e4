"""Is the 10-25 % spread of large materialising launches a property of the
allocation or of what happened just before?  (1) four outputs kept alive, timed in turn, three
rounds; (2) the bench's churn - allocate, time, free (empty_cache) - with and without a pause
after the free, with and without the odd-sized allocation in between."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import fruits_amd as fr
import bench
N, T = 2048, 1024
Xd = bench._device_batch(torch, (N, 3, T), 0)
w15 = fr.words.of_weight(2, dim=3)
plan = fr.ISS([w15[i % 15] for i in range(48)])._plan(0, 48)
plan.prepare(N, T)
K = 48
def t_us(buf):
    return bench._event_time_us(torch, lambda: plan.run(Xd, None, out=buf))
bufs = [torch.empty((K, N, T), dtype=torch.float64, device="cuda") for _ in range(4)]
for rnd in range(3):
    print("alive round", rnd, [round(t_us(b), 1) for b in bufs], flush=True)
del bufs
torch.cuda.empty_cache()
for pause, shift_on in ((0.0, True), (0.5, True), (0.0, False), (0.5, False)):
    ts, shift = [], []
    for trial in range(5):
        buf = torch.empty((K, N, T), dtype=torch.float64, device="cuda")
        ts.append(round(t_us(buf), 1))
        del buf
        torch.cuda.empty_cache()
        if pause:
            time.sleep(pause)
        if shift_on:
            shift.append(torch.empty((trial + 1) * 37_000_001, dtype=torch.uint8, device="cuda"))
    del shift
    torch.cuda.empty_cache()
    print(f"churn pause {pause} shift {shift_on}:", ts, flush=True)
# the same buffer, timed five times with idle gaps of growing length in front
buf = torch.empty((K, N, T), dtype=torch.float64, device="cuda")
for gap in (0.0, 0.05, 0.2, 1.0, 3.0):
    time.sleep(gap)
    print(f"one buffer after {gap} s idle: {t_us(buf):.1f}", flush=True)
