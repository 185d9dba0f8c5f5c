#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo: exercises the multi-rank code path (eager warm-up with bucketed all-reduce from the
# gradient hooks, hipGraph capture, all-reduce between the captured step and the optimizer, max-over-ranks timing) where no
# second GPU is available. RCCL itself is not exercised. Usage: tools/ddp_rehearsal.sh [steps]
set -e
cd "$(dirname "$0")/.."
export TSASR_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-4}
RANK=1 LOCAL_RANK=1 python bench.py --gpus 2 --steps $STEPS --warmup 4 --no-cpu-baseline > /tmp/ddp_rank1.log 2>&1 &
PID1=$!
RANK=0 LOCAL_RANK=0 python bench.py --gpus 2 --steps $STEPS --warmup 4 --no-cpu-baseline
wait $PID1
echo "rank1 exit: $?"; tail -2 /tmp/ddp_rank1.log
