#!/bin/bash
# A/B of HIP runtime knobs on the default bench line (one step = one hipGraph replay of ~730 dependent kernels): usage tools/env_sweep.sh <outdir>
O=${1:-gpurun_out/r3/env}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > $O/$name.json 2> $O/$name.err; echo "$name: $(tail -1 $O/$name.json | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["loss"])' 2>&1 | tail -1)"; }
run base X=1
run devkernarg HIP_FORCE_DEV_KERNARG=1
run nopktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run gq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run gq2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run gq8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run batch16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run batch1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run base2 X=1
