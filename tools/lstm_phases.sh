#!/bin/bash
# Rebuild csrc/lstm.hip with -DLQ_PROFILE in THIS copy of the tree (run on the GPU box: the copy is scratch) and print the phase times
set -e
cd ts-asr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -DLQ_PROFILE -c lstm.hip -o build/lstm.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtsasr_hip.so build/*.o
cd ../..
python tools/lstm_phases.py 1 1921
python tools/lstm_phases.py 32 121
