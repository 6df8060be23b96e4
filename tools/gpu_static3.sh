#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/eq.log
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import fruits_amd as fr
from fruits_amd import _native as nat
words = fr.words.of_weight(2, dim=3)
plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
for N, T in ((2048, 1024), (100, 1000), (7, 600), (3001, 1024)):
    os.environ["FRUITS_HIP_GROUPS"] = "0"
    X = np.random.default_rng(N).standard_normal((N, 3, T))
    Xd = nat.to_device(X)
    os.environ["FRUITS_HIP_STATIC"] = "0"
    ref = torch.full((plan.rows, N, T), float("nan"), dtype=torch.float64, device=Xd.device)
    plan.run(Xd, None, out=ref, groups=1); torch.cuda.synchronize()
    os.environ["FRUITS_HIP_STATIC"] = "1"
    for G in (1, 2, 3):
        for pers in (1, 0):
            os.environ["FRUITS_HIP_PERSIST"] = str(pers)
            out = torch.full((plan.rows, N, T), float("nan"), dtype=torch.float64, device=Xd.device)
            plan.run(Xd, None, out=out, groups=G); torch.cuda.synchronize()
            d = (out - ref).abs()
            print(N, T, "G", G, "persist", pers, "max abs diff vs interpreter", d.max().item(), "finite", bool(torch.isfinite(out).all()))
    os.environ["FRUITS_HIP_PERSIST"] = "1"
PY
V='[{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_PERSIST":1,"FRUITS_HIP_GROUPS":0}'
for G in 1 2 3; do for P in 1 0; do V="$V,{\"FRUITS_HIP_STATIC\":1,\"FRUITS_HIP_PERSIST\":$P,\"FRUITS_HIP_GROUPS\":$G}"; done; done
V="$V]"
for shape in 2048,3,1024 3072,3,1024 8192,3,1024 1024,3,1024 512,3,1024; do
  echo "== shape $shape" | tee -a $O/t.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/t.log
done
