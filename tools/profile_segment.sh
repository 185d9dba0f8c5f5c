#!/bin/bash
# Kernel-by-kernel listing of a window of one replayed step (tools/stream_timeline.py --dump): usage tools/profile_segment.sh <tag> <t0_ms> <t1_ms> [bench args...]
set -e
R=$PWD
TAG=$1; A=$2; B=$3; shift 3
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/stream_timeline.py $db --dump $A $B > $O/segment.txt 2>&1
rm -rf $O/prof
