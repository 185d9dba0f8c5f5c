set -e
R=$PWD
O=$R/gpurun_out/r3/dump
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/stream_timeline.py $db --dump 0 30 > $O/step_dump.txt 2>&1
python tools/trace_summary.py $db --steps 17 --top 100 > $O/trace_summary.txt
rm -rf $O/prof
tail -2 $O/bench_prof.log | cut -c1-400
