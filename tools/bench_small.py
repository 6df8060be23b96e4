"""Latency of small problems (BASELINE configs[0] shape): host overhead matters here."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
X = np.random.default_rng(0).random((200, 3, 100))
def tm(f, n=20):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return f"{np.median(ts)*1e3:.3f} ms"
iss = fr.ISS([fr.words.SimpleWord("[11]")])
print("ISS([11]).fit_transform (200,3,100):", tm(lambda: iss.fit_transform(X)))
iss2 = fr.ISS(fr.words.of_weight(2, 3), mode=fr.ISSMode.EXTENDED)
print("ISS(of_weight(2,3) EXT).fit_transform:", tm(lambda: iss2.fit_transform(X)))
fruit = fr.Fruit("readme")
fruit.add(fr.preparation.INC)
fruit.add(fr.ISS(fr.words.of_weight(2, dim=3), mode=fr.ISSMode.EXTENDED))
fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
print("Fruit.fit:", tm(lambda: fruit.fit(X), 10))
print("Fruit.transform:", tm(lambda: fruit.transform(X)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): fruit.transform(X)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

# a whole experiment fruit on a UCR-sized problem
sys.path.insert(0, "tools")
import bench_pipeline as bp
Xs = np.random.default_rng(1).standard_normal((300, 1, 150)).cumsum(axis=2)
red = bp.build_reduced()
np.random.seed(0)
print("fruit_reduced (300,1,150) fit:", tm(lambda: red.fit(Xs), 5))
print("fruit_reduced (300,1,150) transform:", tm(lambda: red.transform(Xs)))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): red.transform(Xs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
