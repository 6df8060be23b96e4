import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fruits_amd as fr
from fruits_amd import _native as nat
os.environ["FRUITS_HIP_JIT"] = "0"
for word, D in (("[4]", 4), ("[3]", 4), ("[3]", 3), ("[2]", 4), ("[1]", 4), ("[4][4]", 4)):
    for N in (5, 40, 777, 1600):
        for T in (1024, 600):
            X = np.random.default_rng(N).random((N, D, T))
            Xd = nat.to_device(X)
            iss = fr.ISS([fr.words.SimpleWord(word)])
            out = nat.to_host(iss.transform_device(Xd))
            d = int(word[1]) - 1
            if word == "[4][4]":
                c = np.cumsum(X[:, d], axis=1); ref = np.cumsum(X[:, d] * np.concatenate([np.zeros((N, 1)), c[:, :-1]], axis=1), axis=1)
            else:
                ref = np.cumsum(X[:, d, :], axis=1)
            err = np.abs(out[0] - ref).max()
            plan = iss._plan(0, 1)
            print(word, "D", D, "N", N, "T", T, "max err %.3g" % err, "aot", plan.static_program_index(1))
