#!/bin/bash
# Kernel trace of the default bench command (rocprofv3 --kernel-trace --stats; program directly after `--`) summarised per kernel and
# per hardware queue into gpurun_out/<tag>/; usage: tools/profile_step.sh <tag> [bench args...]
set -e
R=$PWD
TAG=${1:-trace}; shift || true
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/trace_summary.py $db --steps 17 --top 80 > $O/trace_summary.txt
python tools/stream_timeline.py $db --names > $O/timeline.txt 2>&1 || true
python tools/stream_timeline.py $db --dump 2.60 2.90 > $O/dump_fwd_layer.txt 2>&1 || true      # one mixture layer forward ...
python tools/stream_timeline.py $db --dump 5.20 5.60 > $O/dump_bwd_layer.txt 2>&1 || true      # ... and backward, kernel by kernel
python tools/stream_timeline.py $db --dump 1.40 2.10 > $O/dump_fwd_beside_lstm.txt 2>&1 || true  # the layers that run beside the persistent LSTM forward
rm -rf $O/prof
tail -4 $O/bench_prof.log | cut -c1-300
