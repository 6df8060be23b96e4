"""`python tools/run_pieces.py cfg4|cfg5 [piece_nodes ...]`: the fused launch of a BASELINE config
on one GPU through the record loop (node shapes as immediates) and IN PIECES of the given sizes
(csrc/plan.h, PiecedProgram) - compile time on a cold cache, time per launch, equality of the
features."""
import os
import sys
import tempfile
import time
import numpy as np
sys.path.insert(0, ".")
import torch
import fruits_amd as fr
from fruits_amd import _native as nat
import bench

which = sys.argv[1]
sets = sys.argv[2:] or ["piece_nodes=64", "piece_nodes=128"]
os.environ["FRUITS_HIP_JIT_CACHE"] = tempfile.mkdtemp(prefix="frjit")
os.environ["FRUITS_AMD_AUTO_PREPARE"] = "0"


def knobs(**kw):
    os.environ["FRUITS_HIP_DEBUG"] = ",".join(f"{k}={v}" for k, v in kw.items())


def build():
    if which == "cfg4":
        return bench._config4(torch, fr, nat)
    if which == "cfg3":
        return bench._Pipeline(torch, fr, nat, (2048, 3, 1024), fr.words.of_weight(4, dim=2),
                               fr.iss.weighting.Indices(), [fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END],
                               n_fit=128)
    return bench._Pipeline(torch, fr, nat, (8192, 6, 4096), fr.words.of_weight(9, dim=1),
                           fr.iss.weighting.L1(), [fr.sieving.NPI, fr.sieving.END], n_fit=32)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


knobs(pieces=0)
p = build()
t0 = time.time()
fn, feats, pipe = p.launch()
print(f"{which} record loop: prepare {time.time() - t0:.1f} s, kernels {pipe.jit_loaded()}", flush=True)
ms = timed(fn)
ref = feats.clone()
print(f"{which} record loop: {ms:.3f} ms per launch", flush=True)
for ks in sets:
    os.environ["FRUITS_HIP_DEBUG"] = "pieces=1,piece_min=32," + ks
    size = int(dict(kv.split("=") for kv in ks.split(",")).get("piece_nodes", 64))
    p.slc._fused_cache = {}          # (a new pipeline: the cover is part of its state)
    p.iss._plans = {}                # (and a new plan: the cover is cached in it)
    t0 = time.time()
    fn, feats, pipe = p.launch()
    dt = time.time() - t0
    cover = pipe.plan.pieces(size)
    ms = timed(fn)
    same = bool((feats == ref).all())
    print(f"{which} [{ks}]: {ms:.3f} ms per launch (equal: {same}); {len(cover['types'])} types "
          f"{[(t['body_nodes'], t['units']) for t in cover['types']]}, chain nodes {cover['chain_nodes']}, "
          f"prepared in {dt:.1f} s, loaded {pipe.pieces_loaded()}", flush=True)
