#!/bin/bash
# Evidence of a round beyond tools/profile_round.sh (trace + PMC passes + default bench): the other bench workloads, the long-form trace,
# phase stamps of an unprofiled replay, and the GEMM micro-benchmarks. usage: tools/evidence_round.sh r04 -> gpurun_out/<tag>/
set -e
R=$PWD
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
for c in ragged accum4 none pretrained longform longform_chunk40; do
  case $c in
    ragged) args="--ragged";; accum4) args="--accum 4";; *) args="--config $c";;
  esac
  timeout -k 10 400 python bench.py --steps 40 --no-cpu-baseline $args > $O/bench_$c.json 2> $O/bench_$c.err || true
  echo "$c: $(grep 'timed region' $O/bench_$c.err)"
done
timeout -k 10 240 python tools/step_stamps.py 3 2>&1 | grep -v Warning > $O/step_stamps.txt || true
timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes.txt || true
timeout -k 10 300 python tools/gemm_bench.py --floors 2>&1 | grep -v amdgpu.ids > $O/gemm_floors.txt || true
timeout -k 10 300 python tools/gemm_bench.py --chain 2>&1 | grep -v amdgpu.ids > $O/gemm_chain.txt || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_lf -o trace -- python3 $R/bench.py --config longform --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_prof_longform.log 2>&1 || true
cd $R
db=$(find $O/prof_lf -name "*.db" | head -1)
python tools/trace_summary.py $db --steps 13 --top 40 > $O/trace_summary_longform.txt || true
rm -rf $O/prof_lf
echo "evidence done"
