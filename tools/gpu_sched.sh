#!/bin/bash
# schedule comparison: product (contiguous spans) vs strided variant, one process per arm and shape
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_GROUPS":1},{"FRUITS_HIP_GROUPS":2},{"FRUITS_HIP_GROUPS":3},{"FRUITS_HIP_GROUPS":6},{"FRUITS_HIP_GROUPS":9}]'
for shape in 64,3,1024 256,3,1024 512,3,1024 768,3,1024 1000,3,1024 1536,3,1024 2048,3,1024 3000,3,1024 4096,3,1024 8192,3,1024; do
  for lib in libfruits_hip.so libfruits_hip.strided.so; do
    echo "== $shape $lib" | tee -a $O/sched.log
    FRUITS_HIP_LIB=$lib TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/sched.log
  done
done
for r in 1 2; do for lib in libfruits_hip.so libfruits_hip.strided.so; do
  FRUITS_HIP_GROUPS=1 FRUITS_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 100 > $O/b_${lib}_$r.json 2> $O/b_${lib}_$r.err || tail -3 $O/b_${lib}_$r.err
  python - <<PY
import json
d=json.load(open("$O/b_${lib}_$r.json")); e=d["extras"]
print("G=1 $lib", "$r", "headline %.1f us" % d["roofline"]["batches"]["median_us"], "w48 %.1f" % e["words48_single"]["kernel_us"],
      "cfg3 %.1f us" % e["config3_fused_pipeline"]["launch_us"], "cfg4 %.2f ms" % (e["config4_single_gpu"]["launch_us"]/1e3),
      "cfg5 %.2f ms" % (e["config5_single_gpu"]["launch_us"]/1e3))
PY
done; done
