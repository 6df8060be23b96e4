#!/bin/bash
# timing-build experiments (the in-tree library must be the FRUITS_HIP_TIMING_BUILD one)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_DEBUG":0},{"FRUITS_HIP_DEBUG":1},{"FRUITS_HIP_DEBUG":2},{"FRUITS_HIP_DEBUG":8},{"FRUITS_HIP_DEBUG":32},{"FRUITS_HIP_DEBUG":33},{"FRUITS_HIP_DEBUG":9},{"FRUITS_HIP_DEBUG":41}]'
for shape in 2048,3,1024 8192,3,1024; do
  echo "== shape $shape" | tee -a $O/dbg.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/dbg.log
done
