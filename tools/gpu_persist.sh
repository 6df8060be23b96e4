#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for r in 1 2; do
  for p in 1 0; do
    FRUITS_HIP_PERSIST=$p python bench.py --no-cpu-baseline --steps 100 > $O/b_${p}_$r.json 2> $O/b_${p}_$r.err || tail -3 $O/b_${p}_$r.err
    python - <<PY
import json
d=json.load(open("$O/b_${p}_$r.json")); e=d["extras"]
print("persist $p", "$r", "headline %.1f us" % d["roofline"]["batches"]["median_us"], "w48 %.1f" % e["words48_single"]["kernel_us"],
      "cfg3 %.1f us" % e["config3_fused_pipeline"]["launch_us"], "cfg4 %.2f ms" % (e["config4_single_gpu"]["launch_us"]/1e3),
      "cfg5 %.2f ms" % (e["config5_single_gpu"]["launch_us"]/1e3))
PY
  done
done
