#!/bin/bash
# same-box A/B against an earlier tree: alternating processes.  _variants/r01 = a checkout of the
# commit to compare with (git worktree add _variants/r01 <commit> && (cd _variants/r01 && python -m
# fruits_amd.build)); git-ignored, not kept in the repository
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
