#!/bin/bash
# static-program walk vs the record interpreter: equality of the outputs, then interleaved timing
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
iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
plan = iss._plan(0, len(words))
for N, T in ((2048, 1024), (100, 1000), (7, 600), (1536, 1024)):
    X = np.random.default_rng(N).standard_normal((N, 3, T))
    Xd = nat.to_device(X)
    outs = []
    for s in (0, 1):
        os.environ["FRUITS_HIP_STATIC"] = str(s)
        out = torch.full((plan.rows, N, T), float("nan"), dtype=torch.float64, device=Xd.device)
        plan.run(Xd, None, out=out, groups=1); torch.cuda.synchronize()
        outs.append(out)
    d = (outs[0] - outs[1]).abs(); rel = (d / outs[0].abs().clamp_min(1e-300)).max().item(); print(N, T, "static == interpreter:", bool(torch.equal(outs[0], outs[1])), "max abs diff", d.max().item(), "max rel", rel, "rows differing", (d.amax(dim=(1,2)) > 0).nonzero().flatten().tolist())
PY
V='[{"FRUITS_HIP_STATIC":0},{"FRUITS_HIP_STATIC":1}]'
for shape in 2048,3,1024 1536,3,1024 8192,3,1024; do
  echo "== shape $shape" | tee -a $O/t.log
  FRUITS_HIP_GROUPS=1 TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/t.log
done
