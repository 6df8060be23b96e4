#!/bin/bash
# occupancy experiment (timing variant): full / no stores / stores without scan at 2,3,4,6 workgroups per CU
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FRUITS_HIP_LIB=libfruits_hip.timing.so FRUITS_HIP_GROUPS=1
V='['
for c in 2 3 4 6; do for d in 0 1 2; do V="$V{\"FRUITS_HIP_PERSIST\":$c,\"FRUITS_HIP_DEBUG\":$d},"; done; done
V="${V%,}]"
for shape in 3072,3,1024 1536,3,1024; do
  echo "== shape $shape" | tee -a $O/occ.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/occ.log
done
