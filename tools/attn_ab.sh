#!/bin/bash
# Per-kernel durations of the attention kernels for each given build of the library (rocprofv3 kernel trace of tools/attn_stamps.py):
#   bash tools/attn_ab.sh OUTDIR B T lib1.so [lib2.so ...]
set -e
R=$PWD
out=$R/$1; B=$2; T=$3; shift 3
mkdir -p "$out"
export TMPDIR=/tmp
for lib in "$@"; do
    name=$(basename "$lib" .so)
    rm -rf "$out/$name"
    rocprofv3 --kernel-trace --stats -d "$out/$name" -o t -- python3 $R/tools/attn_stamps.py "$B" "$T" 0.1 "$R/$lib" > "$out/$name.log" 2>&1
    db=$(find "$out/$name" -name "*.db" | head -1)
    echo "== $name"; grep "fwd .* us" "$out/$name.log" | cut -c1-100
    python3 $R/tools/trace_summary.py "$db" --top 12 | grep -E "attn|dpk" | tee "$out/$name.txt"
    rm -rf "$out/$name"
done
