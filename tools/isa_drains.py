#!/usr/bin/env python3
"""Count the full drains (`s_waitcnt vmcnt(0)`) in the gfx950 ISA of every kernel of csrc/*.hip - the signature of guarded loads and stores:
`if (ptr) load`, `cond ? p[i] : 0`, a store behind a per-row guard each sit in a basic block of their own and are waited for where they
stand, so a run of them is a run of serialized memory round trips (profiles/r03_notes.md section 7). Compiles each file to assembly
(device only, the Makefile's flags) and prints, per kernel with at least `min_drains`: drains, global/buffer loads, MFMAs, loops, lines.
No GPU needed.  usage: python tools/isa_drains.py [min_drains] [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ts-asr_amd", "csrc")
args = sys.argv[1:]
min_drains = int(args.pop(0)) if args and args[0].isdigit() else 3
files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
         "--cuda-device-only", "-S"]
for f in files:
    src = f if os.path.isabs(f) else os.path.join(CSRC, os.path.basename(f))
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", tmp.name], capture_output=True, text=True, cwd=CSRC)
        if r.returncode:
            print(f"{f}: compile failed\n{r.stderr[-400:]}")
            continue
        cur, stats = None, {}
        for ln in open(tmp.name):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                cur = m.group(1)
                stats[cur] = dict(v0=0, loads=0, mfma=0, loops=0, lines=0)
                continue
            if cur is None:
                continue
            s = stats[cur]
            s["lines"] += 1
            s["v0"] += "s_waitcnt" in ln and "vmcnt(0)" in ln
            s["loads"] += "global_load" in ln or "buffer_load" in ln
            s["mfma"] += "v_mfma" in ln
            s["loops"] += "Loop Header" in ln
            if "s_endpgm" in ln:
                cur = None
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["v0"]):
        if s["v0"] >= min_drains:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)[:80]
            print(f"{os.path.basename(f):20s} drains {s['v0']:4d}  loads {s['loads']:4d}  mfma {s['mfma']:3d}  loops {s['loops']:2d}  lines {s['lines']:6d}  {name}")
