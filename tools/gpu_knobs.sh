#!/bin/bash
# the GPU suite under the non-default settings of the main knobs (every path stays tested)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
JITC=$(mktemp -d)   # a private directory: a code object is executable input
for kv in ${KNOBS:-FRUITS_HIP_STATIC=0 FRUITS_HIP_DEBUG=persist=1 FRUITS_HIP_DEBUG=persist=0 FRUITS_HIP_DEBUG=lean=0 FRUITS_HIP_JIT=2 FRUITS_AMD_FUSED_PREP=0 FRUITS_HIP_DEBUG=packed=0 FRUITS_AMD_AUTO_PREPARE=all}; do
  echo "== $kv" | tee -a $O/knobs.log
  # (FRUITS_AMD_AUTO_PREPARE=all: every fused launch through its pipeline's own kernels - the
  # sieves as immediates, small plans as straight-line code: ~4.5 min of suite + compiler)
  env $kv FRUITS_HIP_JIT_CACHE=$JITC python -m pytest tests -q -m gpu -x -k "not jit and not packed and not short_series and not fused_preparation" > $O/pytest_$kv.log 2>&1
  tail -1 $O/pytest_$kv.log | tee -a $O/knobs.log
done
