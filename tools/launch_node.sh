#!/bin/bash
# One node, one process per GPU, RCCL over xGMI -- the exact commands of the multi-GPU configurations.
#
#   tools/launch_node.sh bench   [N] [extra bench.py args]    weak scaling of the headline workload (what the driver runs for N = 1, 2, 4, 8)
#   tools/launch_node.sh config5 [N]                          BASELINE configs[4]: 1024 frame pairs x 128 segments over N ranks
#   tools/launch_node.sh config4 [N]                          BASELINE configs[3]: VOID-shaped depth completion, segments of one image over N ranks
#   (the same control flow on CPU, gloo, world 2 and 8, batch mocked: tests/test_dist_gloo.py::test_bench_multi_rank_control_flow_dry_run)
#
# Per rank: a contiguous share of the host cores (the interpreter, the three host threads of a PairStream and the native schedule loop of
# one rank stay off the other ranks' cores: 8 x ~2 ms of interpreter time per build otherwise land wherever the scheduler puts them),
# OMP_NUM_THREADS sized to that share, HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL's handles).  The binding is done by tools/rank_bind.py,
# which torch.distributed.run starts instead of the script.
set -e
MODE=${1:-bench}; N=${2:-8}; shift || true; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
CORES=$(nproc)
PER=$(( CORES / N )); [ $PER -lt 1 ] && PER=1
export OMP_NUM_THREADS=$(( PER > 16 ? 16 : PER )) SP_CORES_PER_RANK=$PER
PORT=${MASTER_PORT:-29517}
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT tools/rank_bind.py"
case $MODE in
  bench)   exec $RUN bench.py --gpus $N --steps 20 --warmup 5 "$@" ;;
  config5) exec $RUN bench.py --gpus $N --pairs $(( 1024 / N )) --segments 128 --steps 20 --warmup 5 "$@" ;;
  config4) exec $RUN tools/rccl_check.py "$@" ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
