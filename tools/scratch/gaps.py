"""Gaps between consecutive kernels of one hardware queue in the last replayed step of a rocprofv3 kernel-trace CSV (scratch)."""
import csv, glob, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))), key=lambda x: x[0])
ends = [i for i, r in enumerate(rows) if "clip_adamw" in r[2]]
cands = [rows[a + 1: b + 1] for a, b in zip(ends, ends[1:])]
step = min(cands, key=lambda st: st[-1][1] - st[0][0])      # the shortest bracketed step = a replayed one
print("kernels:", len(step), "span %.2f ms" % ((step[-1][1] - step[0][0]) / 1e6))
byq = {}
for r in step: byq.setdefault(r[3], []).append(r)
for q, rs in byq.items():
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(rs, rs[1:])]
    small = [g for g in gaps if -5 < g < 20]
    print(f"queue {q}: {len(rs)} kernels, gaps: median {statistics.median(small):.2f} us, mean of gaps in (-5, 20) us {statistics.mean(small):.2f}, sum of those {sum(small):.0f} us; gaps >= 20 us: {sum(1 for g in gaps if g >= 20)}")
