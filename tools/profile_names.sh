#!/bin/bash
# Kernel census of ONE replayed step (launches and summed duration per kernel name) + a kernel-by-kernel dump of a window with the gap to
# the previous kernel of the same hardware queue: usage tools/profile_names.sh <tag> <t0_ms> <t1_ms> [bench args...]
set -e
R=$PWD
TAG=${1:-names}; A=${2:-5.0}; B=${3:-5.4}; shift 3 || true
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/stream_timeline.py $db --names --dump $A $B > $O/names.txt 2>&1
rm -rf $O/prof
head -5 $O/names.txt
