"""Run-time compiled static program against the interpreter on a plan outside the standard
word sets (interleaved, launches captured into a graph)."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
strs = ["[1][2]", "[12][1]", "[2]", "[1][1][2]", "[3][1]", "[33]", "[2][3][1]", "[13]", "[3][3]", "[1][3]", "[22][1]"]
words = [fr.words.SimpleWord(s) for s in strs]
REPS = 20
for N in (512, 2048, 8192):
    X = np.random.default_rng(0).standard_normal((N, 3, 1024))
    Xd = nat.to_device(X)
    graphs = []
    for jit in (0, 1):
        os.environ["FRUITS_HIP_JIT"] = str(jit)
        plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
        plan.prepare(N, 1024)
        out = torch.empty((plan.rows, N, 1024), dtype=torch.float64, device=Xd.device)
        plan.run(Xd, None, out=out); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(REPS):
                    plan.run(Xd, None, out=out)
        graphs.append((g, plan, out))
    res = {0: [], 1: []}
    for rnd in range(8):
        for i, (g, _, _) in enumerate(graphs):
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            res[i].append(a.elapsed_time(b) / REPS * 1e3)
    K = graphs[0][1].rows
    balg = 8.0 * N * 1024 * (3 + K)
    print(f"N {N:5d} K {K}: interpreter {np.median(res[0]):7.1f} us ({balg / np.median(res[0]) / 8e6:.3f}), "
          f"jit ({graphs[1][1].jit_loaded()} programs) {np.median(res[1]):7.1f} us ({balg / np.median(res[1]) / 8e6:.3f}); "
          f"equal {bool(torch.equal(graphs[0][2], graphs[1][2]))}")
