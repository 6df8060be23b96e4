#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for r in 1 2; do
  for g in 0 2 3; do
    FRUITS_HIP_GROUPS=$g python bench.py --no-cpu-baseline --steps 50 > $O/b_${g}_$r.json 2> $O/b_${g}_$r.err || tail -3 $O/b_${g}_$r.err
    python - <<PY
import json
d=json.load(open("$O/b_${g}_$r.json")); e=d["extras"]
print("groups $g", "$r", "cfg3 %.1f us" % e["config3_fused_pipeline"]["launch_us"], "cfg4 %.2f ms" % (e["config4_single_gpu"]["launch_us"]/1e3),
      "cfg5 %.2f ms" % (e["config5_single_gpu"]["launch_us"]/1e3))
PY
  done
done
