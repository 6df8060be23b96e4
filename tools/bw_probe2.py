import torch, numpy as np
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    ts = []
    for _ in range(7):
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e-3)
    return float(np.median(ts))
n = 2048 * 3 * 1024
x = torch.randn(n, dtype=torch.float64, device="cuda")
out = torch.empty((6, n), dtype=torch.float64, device="cuda")
t = timeit(lambda: out.copy_(x.unsqueeze(0).expand(6, -1)))
print(f"read 50MB + write 302MB (expand copy): {t*1e6:7.1f} us  {(7*n*8)/t/1e9:7.1f} GB/s")
t = timeit(lambda: out.fill_(1.0)); print(f"fill 302MB: {t*1e6:7.1f} us {6*n*8/t/1e9:7.1f} GB/s")
w6 = torch.arange(6, device='cuda', dtype=torch.float64).unsqueeze(1)
t = timeit(lambda: torch.mul(x.unsqueeze(0), w6, out=out))
print(f"read 50MB + write 302MB (broadcast mul): {t*1e6:7.1f} us  {(7*n*8)/t/1e9:7.1f} GB/s")
y = torch.empty(n, dtype=torch.float64, device="cuda")
t = timeit(lambda: y.copy_(x)); print(f"copy 50MB->50MB: {t*1e6:7.1f} us {2*n*8/t/1e9:7.1f} GB/s")
