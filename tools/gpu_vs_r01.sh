#!/bin/bash
# same-box A/B against the round-1 tree (_variants/r01, built there): alternating processes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for r in 1 2 3; do
  for tree in . _variants/r01; do
    ( cd $tree && python bench.py --steps 200 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/b_$(basename $tree)_$r.json 2> /dev/null )
    python - <<PY
import json
d=json.load(open("$O/b_$(basename $tree)_$r.json")); e=d.get("extras",{})
print("$tree", "$r", "headline kernel_avg %.1f us" % d["roofline"]["kernel_avg_us"], "w48 %.1f" % e["words48_single"]["kernel_us"],
      "cfg3 %.1f us" % e["config3_fused_pipeline"]["launch_us"])
PY
  done
done
