#!/bin/bash
# Evidence of a round, one gpurun call: kernel trace + per-queue timeline, FETCH_SIZE / WRITE_SIZE passes, one SQ pass (MFMA busy,
# occupancy, waits) - each a separate rocprofv3 run of the same bench command (program directly after `--`) - then the default bench
# line. usage: tools/profile_round.sh r02   -> gpurun_out/<tag>/ ; copy the summaries into profiles/.
set -e
R=$PWD
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/trace_summary.py $db --steps 17 --top 80 > $O/trace_summary.txt
python tools/stream_timeline.py $db --names > $O/timeline.txt 2>&1 || true
rm -rf $O/prof
echo "trace done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_f.log 2>&1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_w.log 2>&1
echo "write pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $O/pmc_m -o m --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_m.log 2>&1
echo "sq pass done"
cd $R
f=$(find $O/pmc_f -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_w -name "*counter_collection.csv" | head -1); m=$(find $O/pmc_m -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f $w $O/pmc_traffic > /dev/null
python tools/pmc_mfma_summary.py $m $O/pmc_mfma > /dev/null
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # the default bench below reads the traffic of its dominant kernel from here
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
