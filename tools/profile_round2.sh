#!/bin/bash
# usage (on the GPU box): bash tools/profile_round2.sh rNN [targets...]
# per target: rocprofv3 --kernel-trace --stats, then SEPARATE --pmc passes (SQ issue / wait counters,
# FETCH_SIZE, WRITE_SIZE); condensed by tools/profile_summary2.py into gpurun_out/profiles_rNN/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
TARGETS=${@:-cfg2 cfg3 cfg4 cfg5 cos1 cos2}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
# the driver's own command: kernel stats + traffic of the headline
CMD="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- $CMD > $O/bench_trace.log 2>&1 || echo "bench trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- $CMD > $O/bench_fetch.log 2>&1 || echo "bench fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- $CMD > $O/bench_write.log 2>&1 || echo "bench write failed"
for t in $TARGETS; do
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
