#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv): per (kernel, grid) count, average and
share of the traced GPU time; `--last N` keeps only the last N dispatches (e.g. the hipGraph replays of the timed steps)."""
import argparse, collections, csv, re, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--steps", type=int, default=0, help="divide totals by this many steps")
ap.add_argument("--match", default="")
a = ap.parse_args()
rows = []
if a.path.endswith(".db"):
    db = sqlite3.connect(a.path)
    for name, st, en, gx, lds, vg in db.execute("select name, start, end, grid_x, lds_size, vgpr_count from kernels order by start"):
        rows.append((name, st, en, gx, lds, vg))
else:
    for r in csv.DictReader(open(a.path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
st = collections.defaultdict(list)
for name, s, e, gx, lds, vg in rows:
    nm = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    nm = re.sub(r"^void ", "", nm)[:84]
    if a.match and a.match not in nm:
        continue
    st[(nm, gx, lds, vg)].append((e - s) / 1e3)
tot = sum(sum(v) for v in st.values())
div = a.steps or 1
print(f"{len(rows)} dispatches, {tot/1e3:.2f} ms GPU time" + (f", {tot/1e3/div:.2f} ms/step over {div} steps" if a.steps else ""))
for (nm, gx, lds, vg), v in sorted(st.items(), key=lambda kv: -sum(kv[1]))[: a.top]:
    print(f"{sum(v)/tot*100:5.1f}% {sum(v)/1e3/div:7.3f} ms n={len(v)/div:7.1f} avg={sum(v)/len(v):8.1f}us grid={gx:>8} lds={lds!s:>6} vgpr={vg!s:>4} {nm}")
