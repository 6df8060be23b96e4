"""Debug helper: the materialising walk with and without the lean kernel on one random case."""
import os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import fruits_amd as fr
import test_hip_parity as t
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(1000 + seed)
D = int(rng.integers(1, 5))
N = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 24, 40]))
T = int(rng.choice([1, 2, 3, 17, 64, 129, 256, 300, 385, 513, 700, 1025, 1500]))
words = [t._random_word(rng, D) for _ in range(int(rng.integers(1, 13)))]
if rng.random() < 0.5:
    words += [words[0], words[-1]]
X = np.random.default_rng(3).random((N, D, T)) * 0.9 + 0.3
outs = {}
for lean in (1, 0):
    os.environ["FRUITS_HIP_DEBUG"] = f"lean={lean}"
    iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED)
    outs[lean] = iss.fit_transform(X)
a, b = outs[1], outs[0]
print(words, a.shape)
plan = iss._plan(0, len(words))
for k in range(a.shape[0]):
    bad = ~np.isclose(a[k], b[k], rtol=1e-9, atol=0)
    if bad.any():
        n, tt = np.argwhere(bad)[0]
        print("row", k, "bad", int(bad.sum()), "first at series", n, "t", tt, a[k, n, tt], b[k, n, tt],
              "bad t range", np.argwhere(bad)[:, 1].min(), np.argwhere(bad)[:, 1].max())
