#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh rNN
# kernel-trace stats and, in SEPARATE passes, the HBM byte counters of bench.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
O=$R/gpurun_out/prof_$TAG
rm -rf $O
mkdir -p $O
cd $R
CMD="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1 || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1 || echo "write failed"
python3 tools/profile_summary.py $O $TAG
