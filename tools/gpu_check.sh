#!/bin/bash
# usage (on the GPU box): bash tools/gpu_check.sh [tag]  - GPU test suite, bench N=1, and the
# rehearsals of the N>1 bench path that fit a one-GPU box (nccl world 1; gloo world 2)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-check}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py > $O/bench1.json 2> $O/bench1.err || { tail -20 $O/bench1.err; exit 1; }
FRUITS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
  python bench.py --steps 50 > $O/bench_nccl1.json 2> $O/bench_nccl1.err || { tail -20 $O/bench_nccl1.err; exit 1; }
FRUITS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 50 --warmup 5 \
  > $O/bench_gloo2.json 2> $O/bench_gloo2.err || { tail -20 $O/bench_gloo2.err; exit 1; }
echo done
