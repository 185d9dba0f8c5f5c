#!/bin/bash
# Kernel trace of tools/rccl_single_rank_check.py (real RCCL collectives captured in the step's hipGraph, one-rank communicator):
# which RCCL kernels ran, on which hardware queue, beside what (on a ONE-rank communicator RCCL's all-reduce is its `oneRankReduce` kernel). usage: tools/profile_rccl.sh <tag> -> gpurun_out/<tag>/rccl_*.txt
set -e
R=$PWD
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/prof_rccl -o trace -- python3 $R/tools/rccl_single_rank_check.py > $O/rccl_check.log 2>&1
cd $R
db=$(find $O/prof_rccl -name "*.db" | head -1)
python tools/trace_summary.py $db --steps 1 --top 400 | grep -i -E "nccl|rccl|oneRankReduce|dispatches" > $O/rccl_kernels.txt || true
python tools/stream_timeline.py $db 13 --grep onerankreduce > $O/rccl_timeline.txt 2>&1 || true   # step 13 = a replay of the fp32-payload run
rm -rf $O/prof_rccl
tail -3 $O/rccl_check.log
