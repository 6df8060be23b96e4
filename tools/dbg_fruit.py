import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
import fruits_amd as fr
from test_hip_parity import build_fruit
from oracle import ref_numpy as orc
G = load_golden()
name = sys.argv[1]
case = [c for c in G.cases("fruit") if c["name"] == name][0]
X = G[case["x"]]
fruit = build_fruit(fr, case["spec"])
np.random.seed(case["np_seed"]); fruit.fit(X)
out = fruit.transform(X); ref = G[case["out"]]
d = np.abs(out - ref) / np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-300)
bad = np.argwhere(d > 1e-6)
print(len(bad), "bad entries")
for n, c in bad[:20]:
    print(n, c, case["labels"][c], out[n, c], ref[n, c])
# compare fitted quantiles with the oracle's
fitted = orc.fruit_fit(case["spec"], X, case["np_seed"])
sv_o = fitted[0][1]
sv_m = fruit.get_slice(0)._sieves_extended
for i in range(len(sv_m)):
    for a, b in zip(sv_m[i], sv_o[i]):
        if hasattr(a, "_quantiles") and b.quantiles is not None:
            dq = np.abs(np.asarray(a._quantiles) - b.quantiles)
            dq = dq[np.isfinite(dq)]
            if dq.size and dq.max() > 1e-9 * max(1, np.abs(b.quantiles[np.isfinite(b.quantiles)]).max()):
                print("quantile mismatch", i, a, a._quantiles, b.quantiles)
