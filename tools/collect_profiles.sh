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
ls -la profiles/${TAG}_*
