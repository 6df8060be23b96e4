"""experiments/fruit_reduced.py slice 1 ((2048,3,1024): NEW(INC) -> STD -> ISS(of_weight(4,2), Indices) ->
3 NPI + 3 MPI + END) as ONE straight-line plan (115 nodes x 4 ops) and IN PIECES of <= 64 nodes:
prepare time on a cold cache and time per fused launch."""
import os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, torch
import fruits_amd as fr
import bench
os.environ["FRUITS_HIP_JIT_CACHE"] = tempfile.mkdtemp(prefix="frjit")
os.environ["FRUITS_HIP_JIT_BUNDLE"] = ""
os.environ["FRUITS_AMD_AUTO_PREPARE"] = "0"
N, D, T = 2048, 3, 1024
X = np.random.default_rng(0).standard_normal((N, D, T)).cumsum(axis=2) / 8
S = fr.sieving
def build():
    fruit = fr.Fruit("reduced 1")
    fruit.add(fr.preparation.NEW(fr.preparation.INC()), fr.preparation.STD)
    fruit.add(fr.ISS(fr.words.of_weight(4, 2), mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices()))
    fruit.add(S.NPI(q=(0.5, 1.0), inc=0), S.NPI(q=(0.5, 1.0), inc=1), S.NPI(q=(0.5, 1.0), inc=2),
              S.MPI(q=(0.5, 1.0), inc=0), S.MPI(q=(0.5, 1.0), inc=1), S.MPI(q=(0.5, 1.0), inc=2), S.END)
    fruit.get_slice().fit_sample_size = 0.1
    np.random.seed(0)
    fruit.fit(X)
    return fruit
ref = None
for name, knobs in (("one straight-line plan", "piece_min=200"), ("pieces <= 64", ""), ("pieces <= 128", "piece_nodes=128"),
                    ("generic", None)):
    os.environ["FRUITS_HIP_DEBUG"] = knobs or ""
    fruit = build()
    slc = fruit.get_slice()
    cache = fr.cache.SharedSeedCache(X)
    cache.input_device(X)
    t0 = time.time()
    pipe = slc._fused(T)
    chain = slc._fusable_preparation(T)
    assert pipe.set_preparation(D, *chain)
    if knobs is not None:
        pipe.prepare(N)
    dt = time.time() - t0
    out = slc.transform_device(X, cache=cache)
    torch.cuda.synchronize()
    us = bench._event_time_us(torch, lambda: slc.transform_device(X, cache=cache), reps=10)
    if ref is None:
        ref = out.clone()
    lab = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = torch.tensor(["MPI" not in l for l in lab], device="cuda")
    same = bool((out[:, exact] == ref[:, exact]).all()) and bool(torch.allclose(out, ref, rtol=1e-12, atol=1e-300, equal_nan=True))
    print(f"{name}: prepared in {dt:.1f} s, {us:.1f} us per transform_device, static {pipe.jit_loaded(static_only=True)} "
          f"pieces {pipe.pieces_loaded()} own {pipe.jit_loaded()}, equal {same}", flush=True)
