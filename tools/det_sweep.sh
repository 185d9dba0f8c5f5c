#!/bin/bash
# Determinism sweep, one gpurun call: tools/det_stress.py under the knobs that discriminate stream forks / pools / runtime graph paths.
# usage: tools/det_sweep.sh <outdir> [reps]
O=${1:-gpurun_out/r3/det}; REPS=${2:-15}
mkdir -p $O
run() { # name, reps, env..., -- args
  name=$1; shift; reps=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" ; env "${envs[@]}" timeout -k 10 560 python tools/det_stress.py --reps $reps "$@" > $O/$name.log 2>&1
  echo "rc=$?"; grep "^rep\|RESULT\|poisoned\|Error" $O/$name.log | cut -c1-1200 | tail -12
}
run default $REPS X=1 -- --snap 0 --steps 8 --modes graph
run overlap2 $REPS TSASR_OVERLAP=2 -- --snap 0 --steps 8 --modes graph
run eager_accum2 $((REPS/4)) X=1 -- --snap 0 --steps 8 --accum 2 --modes eager
true
