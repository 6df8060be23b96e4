"""CosWISS measurement: the fruit_reduced CosWISS slice shape
(experiments/fruit_reduced.py:52-69) on synthetic input; device time per stage."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
from fruits_amd import _native as nat

N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 1024
X = np.random.default_rng(0).standard_normal((N, 1, T))
words = fr.words.of_weight(1, 2) + fr.words.of_weight(2, 2) + fr.words.of_weight(3, 2)
for e in (1, 2):
    cw = fr.CosWISS(words, [i / 20 for i in range(1, 11, 2)], exponent=e, total_weighting=True)
    Xd = nat.to_device(np.concatenate([X, np.diff(X, axis=2, prepend=0)], axis=1))
    out = cw.transform_device(Xd)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    ev0.record()
    for _ in range(reps):
        cw.transform_device(Xd, out=out)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    terms = sum(cw._n_terms(w) for w in range(len(words))) * len(cw._freqs)
    rows = cw.n_iterated_sums()
    scans = sum((e + 1) * len(w) for w in words) * len(cw._freqs)
    gb = rows * N * T * 8 / 1e9
    print(f"exponent {e}: {len(words)} words x 5 freqs = {rows} rows ({terms} reference terms, "
          f"{scans} scans), {ms:.3f} ms/transform, {gb / ms:.2f} TB/s written, "
          f"{scans * N * T / ms / 1e6:.1f} G scan-elem/s", flush=True)
    t0 = time.time()
    fruit = fr.Fruit()
    fruit.add(fr.preparation.NEW(fr.preparation.INC()), fr.preparation.STD, cw._copy())
    for k in ("NPI", "MPI"):
        for inc in (0, 1, 2):
            fruit.add(getattr(fr.sieving, k)(q=(0.5, 1.0), inc=inc))
    fruit.add(fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(0)
    fruit.fit(X)
    torch.cuda.synchronize()
    t1 = time.time()
    F = fruit.transform(X)
    torch.cuda.synchronize()
    t2 = time.time()
    F = fruit.transform(X)
    torch.cuda.synchronize()
    t3 = time.time()
    print(f"  slice: fit {t1 - t0:.3f} s, transform {t2 - t1:.3f} s (2nd {t3 - t2:.3f} s), "
          f"features {F.shape}", flush=True)
