#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_GROUPS":1},{"FRUITS_HIP_GROUPS":2},{"FRUITS_HIP_GROUPS":3},{"FRUITS_HIP_GROUPS":6}]'
for shape in 1024,3,1024 2048,3,1024 4096,3,1024; do
  for lib in libfruits_hip.so libfruits_hip.strided.so; do
    echo "== w48 $shape $lib" | tee -a $O/sched48.log
    TUNE_TILE48=1 FRUITS_HIP_LIB=$lib TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/sched48.log
  done
done
