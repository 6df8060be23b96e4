import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
import fruits_amd as fr
from test_hip_parity import make_iss
G = load_golden()
for name in ["arctic_neg_words", "arctic_idx_total", "arctic_l1_G", "arctic_idx_nontotal"]:
    case = [c for c in G.cases("iss") if c["name"] == name][0]
    X = G.x_of(case)
    out = make_iss(fr, case).fit_transform(X)
    ref = G[case["out"]]
    d = np.abs(out - ref)
    k, n, t = np.unravel_index(np.argmax(d), d.shape)
    print(name, "max abs diff", d.max(), "at", (k, n, t), out[k, n, t], ref[k, n, t], "rows with diff:", sorted(set(np.argwhere(d > 0)[:, 0].tolist())))
