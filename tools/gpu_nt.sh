#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_NT_INPUT":0},{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_NT_INPUT":1},{"FRUITS_HIP_STATIC":1,"FRUITS_HIP_NT_INPUT":1}]'
for shape in 2048,3,1024 1024,3,1024 1536,3,1024 2560,3,1024 3072,3,1024; do
  echo "== shape $shape" | tee -a $O/t.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | sed 's/FRUITS_HIP_//g' | tee -a $O/t.log
done
echo "== of_weight(3,2) (26 words, interpreter only), 1024 series" | tee -a $O/t.log
TUNE_WORDS=3,2 TUNE_SHAPE=1024,2,1024 python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | sed 's/FRUITS_HIP_//g' | tee -a $O/t.log
