#!/bin/bash
# same-box comparison of experiment libraries (FRUITS_HIP_LIB) and the round-1 tree on the headline launch
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
shift
V='[{"FRUITS_HIP_GROUPS":0},{"FRUITS_HIP_GROUPS":1}]'
for r in 1 2; do
  for lib in "$@"; do
    echo "== $lib round $r" | tee -a $O/bisect.log
    FRUITS_HIP_LIB=$lib python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/bisect.log
  done
  echo "== r01 tree round $r" | tee -a $O/bisect.log
  ( cd _variants/r01 && python tools/tune2.py '[{"FRUITS_HIP_GROUPS":0}]' 2>&1 | grep -v amdgpu.ids ) | tee -a $O/bisect.log
done
