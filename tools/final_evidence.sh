set -e
mkdir -p gpurun_out/r3
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t_final2.log 2>&1 || { tail -30 gpurun_out/r3/t_final2.log; exit 1; }
tail -2 gpurun_out/r3/t_final2.log
bash tools/profile_round.sh r03
bash tools/r3_lines.sh gpurun_out/r3/lines
timeout -k 10 240 python tools/step_stamps.py 3 2>&1 | grep -v Warning > gpurun_out/r3/stamps_final.txt || true
timeout -k 10 120 python tools/replay_host_time.py 20 2>&1 | grep -v Warning > gpurun_out/r3/host_time.txt || true
tail -3 gpurun_out/r3/host_time.txt
