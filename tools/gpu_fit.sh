#!/bin/bash
# fit: parity of the device-side selection, then timing of fruit_reduced fit / transform
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "select_ranks or device_fit or fruit" > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/bench_pipeline.py reduced 2>&1 | grep -v amdgpu.ids | tee $O/reduced.log
