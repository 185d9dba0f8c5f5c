#!/bin/bash
# The bench lines of the round besides the default one: usage tools/r3_lines.sh <outdir>
O=${1:-gpurun_out/r3/lines}; mkdir -p $O
for cfg in "ragged --ragged" "accum4 --accum 4" "none --config none" "pretrained --config pretrained" "longform --config longform"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 280 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err
  echo "$name rc=$? $(tail -1 $O/$name.json | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], "ms", d["value"], d["unit"], d["config"]["workload"][:60])' 2>&1 | tail -1)"
done
