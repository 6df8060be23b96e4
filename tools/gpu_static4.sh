#!/bin/bash
# static programs: groups x persistence x non-temporal input, against the interpreter
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V='[{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_GROUPS":0,"FRUITS_HIP_NT_INPUT":0,"FRUITS_HIP_STATIC_PERSIST":0},{"FRUITS_HIP_STATIC":0,"FRUITS_HIP_GROUPS":1,"FRUITS_HIP_NT_INPUT":1,"FRUITS_HIP_STATIC_PERSIST":0}'
for G in 1 3; do for P in 0 1; do for NT in 0 1; do
  V="$V,{\"FRUITS_HIP_STATIC\":1,\"FRUITS_HIP_GROUPS\":$G,\"FRUITS_HIP_NT_INPUT\":$NT,\"FRUITS_HIP_STATIC_PERSIST\":$P}"
done; done; done
V="$V]"
for shape in ${SHAPES:-2048,3,1024 1024,3,1024 1536,3,1024 3072,3,1024 8192,3,1024 512,3,1024}; do
  echo "== shape $shape" | tee -a $O/t.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | sed 's/FRUITS_HIP_//g' | tee -a $O/t.log
done
