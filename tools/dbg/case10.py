import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fruits_amd as fr
from test_hip_parity import build_fruit, oracle_features
rng = np.random.default_rng(5010)
N, D, T = 11, 1, 513
X = rng.standard_normal((N, D, T)).cumsum(axis=2) / np.sqrt(T)
words = ['[111]', '[11]', '[11][11]', '[1][111][111][1][1]', '[1][11]', '[1][1][111][11][111]']
for semiring in ("Arctic", "Reals"):
    for weighting in (None, {"kind": "L1", "scale": 2.0}, {"kind": "Indices", "scale": 2.0}, {"kind": "L1", "scale": 2.0, "total": True}):
        for TT in (513, 300):
            spec = {"slices": [{"preps": [{"kind": "INC"}],
                                "iss": [{"words": words, "mode": "SINGLE", "semiring": semiring, "weighting": weighting}],
                                "sieves": [{"kind": "END", "cut": [85, -1]}], "fit_sample_size": 1.0}]}
            Xs = X[:, :, :TT]
            fruit = build_fruit(fr, spec)
            np.random.seed(1)
            fruit.fit(Xs)
            got = fruit.transform(Xs)
            ref, expo = oracle_features(spec, Xs, Xs, np_seed=1)
            bad = np.abs(got - ref) > 1e-6 * (1 + np.abs(ref))
            print(semiring, weighting, TT, "bad cols:", sorted(set(np.nonzero(bad)[1].tolist())))
