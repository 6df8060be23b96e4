#!/bin/bash
# usage (on the GPU box): bash tools/gpu_ab.sh TAG [targets...]  - same-box A/B of two builds of the
# library: fruits_amd/libfruits_hip.prev.so (built from the commit to compare with) against the
# tree's, alternating processes (boxes differ by 5-15 %: only such comparisons count)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
TARGETS=${@:-cfg3 cfg4 cfg5}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2 3; do
  for lib in libfruits_hip.prev.so libfruits_hip.so; do
    for t in $TARGETS; do
      reps=20; [ $t = cfg4 ] && reps=4; [ $t = cfg5 ] && reps=3
      echo -n "$lib " | tee -a $O/ab.txt
      FRUITS_HIP_LIB=$lib python tools/run_kernels.py $t $reps 2>&1 | tail -1 | tee -a $O/ab.txt
    done
  done
done
