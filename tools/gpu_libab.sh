#!/bin/bash
# usage: bash tools/gpu_libab.sh <tag> <other lib>: the in-tree library against another build of it,
# alternating processes, the static-program path on the headline family of shapes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "static or jit" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
V='[{"FRUITS_HIP_STATIC":1}]'
for r in 1 2; do
  for shape in 2048,3,1024 1536,3,1024 3072,3,1024 8192,3,1024 512,3,1024; do
    for lib in libfruits_hip.so $2; do
      echo -n "$lib $shape: " | tee -a $O/t.log
      FRUITS_HIP_LIB=$lib TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep median | sed 's/.*median/median/' | tee -a $O/t.log
    done
  done
done
