"""`python tools/bench_argmax.py [N] [T]`: a slice ISS(of_weight(3, 2), EXTENDED, Arctic(argmax=True))
-> NPI / NPI(inc=1) / MPI / END: FruitSlice.transform with the argmax rows formed in LDS
(fr_pipeline_set_argmax) against the materialising path (fr_arctic_argmax + a launch per row and
sieve), same thresholds; ms per call, features compared."""
import os, sys, time, numpy as np
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X = np.random.default_rng(0).standard_normal((N, 2, T)).cumsum(axis=2)
fruit = fr.Fruit()
fruit.add(fr.ISS(fr.words.of_weight(3, 2), mode=fr.ISSMode.EXTENDED, semiring=fr.semiring.Arctic(argmax=True)))
fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.NPI(q=(0.5, 1.0), inc=1), fr.sieving.MPI(q=(0.5, 1.0)),
          fr.sieving.END)
np.random.seed(0)
fruit.fit(X[:128])
slc = fruit.get_slice()
out = {}
for mode in ("1", "0"):
    os.environ["FRUITS_AMD_FUSED_ARGMAX"] = mode
    slc._fused_cache = {}
    fruit.transform(X)
    ts = []
    for _ in range(5 if mode == "1" else 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out[mode] = fruit.transform(X)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"argmax rows in LDS = {mode}: fused pipeline {slc._fused(T) is not None}, "
          f"{min(ts) * 1e3:.1f} ms per transform, {slc.niteratedsums()} rows, {out[mode].shape[1]} features")
mpi = np.array([type(sv).__name__ == "MPI" for sv in slc._sieves for _ in range(sv.nfeatures())] * slc.niteratedsums())
print("counts / values equal:", bool(np.array_equal(out["1"][:, ~mpi], out["0"][:, ~mpi])),
      " band means max rel diff:", float(np.max(np.abs(out["1"][:, mpi] - out["0"][:, mpi]) / (np.abs(out["0"][:, mpi]) + 1e-300))))
