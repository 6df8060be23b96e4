import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fruits_amd as fr
from test_hip_parity import build_fruit
from oracle import ref_numpy as orc
T = int(sys.argv[1])
rng = np.random.default_rng(T)
X = rng.standard_normal((3, 2, T))
spec = {"slices": [{"iss": [{"words": ["[1]", "[12]"], "mode": "SINGLE"}],
                    "sieves": [{"kind": "NPI", "inc": 2}, {"kind": "MPI", "inc": 2}, {"kind": "NPI", "inc": 1}]}]}
fruit = build_fruit(fr, spec); fruit.fit(X)
got = fruit.transform(X)
ref = orc.fruit_transform(spec, orc.fruit_fit(spec, X), X)
os.environ["FRUITS_AMD_FUSED"] = "0"
for s in fruit._slices: s._fused_cache = {}
plain = fruit.transform(X)
print("fused\n", got, "\nunfused\n", plain, "\noracle\n", ref)
