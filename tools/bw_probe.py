import torch, numpy as np
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e-3)
    return float(np.median(ts))
for mb in [302, 805, 2000]:
    n = mb * 1000 * 1000 // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    t = timeit(lambda: x.fill_(1.0)); print(f"fill  {mb} MB: {t*1e6:8.1f} us  {n*8/t/1e9:8.1f} GB/s written")
    t = timeit(lambda: x.zero_());    print(f"zero  {mb} MB: {t*1e6:8.1f} us  {n*8/t/1e9:8.1f} GB/s written")
    t = timeit(lambda: y.copy_(x));   print(f"copy  {mb} MB: {t*1e6:8.1f} us  {2*n*8/t/1e9:8.1f} GB/s read+written")
    t = timeit(lambda: x.sum());      print(f"sum   {mb} MB: {t*1e6:8.1f} us  {n*8/t/1e9:8.1f} GB/s read")
    t = timeit(lambda: torch.add(x, 1.0, out=y)); print(f"add   {mb} MB: {t*1e6:8.1f} us  {2*n*8/t/1e9:8.1f} GB/s r+w")
