#!/bin/bash
# usage: bash tools/gpu_ab.sh <tag> ; parity suite first, then A/B of group choices (one process each shape)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
V='[{"FRUITS_HIP_GROUPS":0},{"FRUITS_HIP_GROUPS":1},{"FRUITS_HIP_GROUPS":2},{"FRUITS_HIP_GROUPS":3},{"FRUITS_HIP_GROUPS":6},{"FRUITS_HIP_GROUPS":9},{"FRUITS_HIP_GROUPS":0,"FRUITS_HIP_PREFETCH":0}]'
for shape in 2048,3,1024 1536,3,1024 4096,3,1024 1000,3,1024; do
  echo "== shape $shape" | tee -a $O/ab.log
  TUNE_SHAPE=$shape python tools/tune2.py "$V" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
python bench.py --quick-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['roofline']['batches'], d['extras']['words48_single'], d['extras']['config3_fused_pipeline']['launch_us'])"
