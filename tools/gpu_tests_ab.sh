#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; rc=$?
tail -30 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
V='[{"FRUITS_HIP_GROUPS":0},{"FRUITS_HIP_GROUPS":1},{"FRUITS_HIP_GROUPS":2},{"FRUITS_HIP_GROUPS":3},{"FRUITS_HIP_GROUPS":6},{"FRUITS_HIP_GROUPS":9}]'
for shape in 64,3,1024 256,3,1024 512,3,1024 768,3,1024 1000,3,1024 3000,3,1024; do
  echo "== shape $shape" | tee -a $O/ab.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
