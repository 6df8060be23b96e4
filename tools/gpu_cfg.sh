#!/bin/bash
# usage (on the GPU box): bash tools/gpu_cfg.sh [tag] [notest]  - GPU suite, then the fused launches of
# BASELINE configs 3 / 4 / 5 (tools/run_kernels.py) timed back to back
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-cfg}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then
  python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
  tail -5 $O/pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
for t in cfg3 cfg4 cfg5; do
  reps=20; [ $t = cfg4 ] && reps=5; [ $t = cfg5 ] && reps=3
  python tools/run_kernels.py $t $reps 2>&1 | tail -1 | tee -a $O/times.txt
done
