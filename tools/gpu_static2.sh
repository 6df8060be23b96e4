#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_PERSIST":1},{"FRUITS_HIP_STATIC":1,"FRUITS_HIP_PERSIST":1},{"FRUITS_HIP_STATIC":1,"FRUITS_HIP_PERSIST":0},{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_PERSIST":0}]'
for shape in 2048,3,1024 1536,3,1024 3072,3,1024 8192,3,1024; do
  echo "== shape $shape" | tee -a $O/t.log
  FRUITS_HIP_GROUPS=1 TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/t.log
done
