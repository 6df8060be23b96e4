"""`python tools/placement_probe.py [w48|k115] [buffers] [launches]`: the same materialising launch on
`buffers` fresh allocations of its output, all kept alive (so every one lies somewhere else),
`launches` launches each, a torch fill_ of the buffer after them; prints per buffer the address and
the times.  Under `rocprofv3 --pmc ...` the counter file holds one row per dispatch in the same
order: tools/placement_counters.py groups them by buffer (where the output lies
moves these launches by 10-25 %)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import fruits_amd as fr
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "w48"
n_buf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 6
N, T = 2048, 1024
Xd = bench._device_batch(torch, (N, 3, T), 0)
if which == "w48":
    w15 = fr.words.of_weight(2, dim=3)
    iss = fr.ISS([w15[i % 15] for i in range(48)])
else:
    iss = fr.ISS(fr.words.of_weight(4, dim=2), mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(iss.words))
plan.prepare(N, T)
K = plan.rows
warm = torch.empty((K, N, T), dtype=torch.float64, device="cuda")
for _ in range(5):
    plan.run(Xd, None, out=warm)
torch.cuda.synchronize()
print(f"PROBE {which} K {K} warm-up launches 5 buffers {n_buf} launches {n_launch}", flush=True)
bufs, shift = [], []
for b in range(n_buf):
    shift.append(torch.empty((b + 1) * 37_000_001, dtype=torch.uint8, device="cuda"))
    buf = torch.empty((K, N, T), dtype=torch.float64, device="cuda")
    bufs.append(buf)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch + 1)]
    ev[0].record()
    for i in range(n_launch):
        plan.run(Xd, None, out=buf)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n_launch)]
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    buf.fill_(1.0)
    c.record()
    torch.cuda.synchronize()
    p = buf.data_ptr()
    print(f"BUF {b} ptr {p:#x} mod2M {p % (1 << 21):#x} mod1G {(p % (1 << 30)) >> 21} x2M  "
          f"launch us {np.median(ts):.1f} (min {min(ts):.1f})  fill us {a.elapsed_time(c) * 1e3:.1f}", flush=True)
