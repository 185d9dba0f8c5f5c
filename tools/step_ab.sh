#!/bin/bash
# Same-lease A/B of the whole training step between the package's library and another build of it (e.g. the previous commit's, linked
# by hand as lib/libtsasr_prev.so): the two are swapped in place on the GPU box's copy of the tree and bench.py runs alternately.
#   bash tools/step_ab.sh lib/libtsasr_prev.so [rounds] [bench args...]
set -e
L=ts-asr_amd/lib/libtsasr_hip.so
other=ts-asr_amd/$1; rounds=${2:-3}; shift 2 || true
cp $L /tmp/step_ab_new.so
P='import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])'
for i in $(seq $rounds); do
    cp /tmp/step_ab_new.so $L; echo -n "new   "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "$P"
    cp $other $L;              echo -n "other "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "$P"
done
cp /tmp/step_ab_new.so $L
