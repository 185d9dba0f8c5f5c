#!/bin/bash
# Copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ (tracked): usage tools/collect_profiles.sh r02
set -e
TAG=${1:-r02}
O=gpurun_out/$TAG
{
  echo "# $TAG kernel trace: rocprofv3 --kernel-trace --stats of \`python bench.py --steps 10 --warmup 3 --no-cpu-baseline\`"
  echo
  echo "Per-kernel rows are normalised to ONE training step (the trace holds 17: 3 warm-up + 1 capture pass + 3 instrumented eager + 10 timed"
  echo "replays); n = launches per step, avg = mean rocprofv3 duration of a launch. Summary by tools/trace_summary.py, per-queue timeline"
  echo "of one replayed step by tools/stream_timeline.py."
  echo
  echo '```'
  cat $O/trace_summary.txt
  echo '```'
  echo
  echo "## One replayed step, per hardware queue"
  echo
  echo '```'
  cat $O/timeline.txt
  echo '```'
} > profiles/${TAG}_kernel_trace.md
cp $O/pmc_traffic.md profiles/${TAG}_pmc_traffic.md
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp $O/pmc_mfma.md profiles/${TAG}_pmc_mfma.md
cp $O/pmc_mfma.json profiles/${TAG}_pmc_mfma.json
tail -1 $O/bench.json > profiles/${TAG}_bench.json
for c in pretrained longform longform_chunk40; do
  if [ -s $O/bench_$c.json ]; then tail -1 $O/bench_$c.json > profiles/${TAG}_bench_$c.json; fi
done
if [ -s $O/rccl_timeline.txt ]; then
  {
    echo "# $TAG: real RCCL collectives captured in the step's hipGraph (ONE-rank communicator, one GPU)"
    echo
    echo "rocprofv3 --kernel-trace of \`tests/helpers/rccl_single_rank_check.py\`: the gradient arena is told there are two ranks, so"
    echo "every bucket all-reduce of the step is issued through csrc/comm.hip (bare ncclAllReduce on the communication stream), captured into"
    echo "the hipGraph with the rest of the step and replayed. On a one-rank communicator RCCL's all-reduce kernel is \`oneRankReduce\`. This"
    echo "shows capture + replay + placement beside backward; it does NOT validate N > 1 (no multi-GPU box in the development loop)."
    echo
    echo "Output of the check (losses of the plain / RCCL fp32 payload / RCCL bf16 payload / collectives-outside-the-graph runs, bucket collectives in the captured step):"
    echo '```'
    grep -E "^(plain|rccl|rccl_bf16|rccl_uncaptured) |RCCL single-rank" $O/rccl_check.log | cut -c1-260
    echo '```'
    echo
    echo "RCCL kernels in the whole trace (3 runs x 8 steps):"
    echo '```'
    cat $O/rccl_kernels.txt
    echo '```'
    echo
    echo "One replayed step of the fp32-payload run, per hardware queue, and every RCCL kernel in it with what ran beside it:"
    echo '```'
    cat $O/rccl_timeline.txt
    echo '```'
  } > profiles/${TAG}_rccl_single_rank_trace.md
fi
ls -la profiles/${TAG}_*
