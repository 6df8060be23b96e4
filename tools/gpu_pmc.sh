#!/bin/bash
# usage (on the GPU box): bash tools/gpu_pmc.sh TAG target [target...]   - per target (tools/run_kernels.py):
# rocprofv3 --kernel-trace --stats, then separate --pmc passes; condensed by tools/profile_summary2.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
for t in "$@"; do
  reps=10; [ $t = cfg4 ] && reps=3; [ $t = cfg5 ] && reps=2
  RUN="python3 tools/run_kernels.py $t $reps"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${t}_trace -- $RUN > $O/${t}_trace.log 2>&1 || echo "$t trace failed"
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
             "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/${t}_pmc$i -- $RUN > $O/${t}_pmc$i.log 2>&1 || echo "$t pmc pass $i failed"
  done
  echo "profiled $t"
done
python3 tools/profile_summary2.py $O $TAG
