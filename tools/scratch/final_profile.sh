# final evidence of a round: kernel trace + per-queue timeline, FETCH_SIZE / WRITE_SIZE passes (separate), default bench line
set -e
R=$PWD
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/trace_summary.py $db --steps 17 --top 70 > $O/trace_summary.txt
python tools/stream_timeline.py $db > $O/timeline.txt 2>&1 || true
echo "trace done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_f.log 2>&1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_w.log 2>&1
echo "write pass done"
cd $R
f=$(find $O/pmc_f -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_w -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f $w $O/pmc_traffic
rm -rf $O/prof $O/pmc_f $O/pmc_w
cp $O/pmc_traffic.json profiles/r01_pmc_traffic.json   # the default bench below reads the traffic of its dominant kernel from here
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
