"""Median duration per (kernel, grid) from a rocprofv3 kernel trace CSV; substring filters as arguments (scratch)."""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[-1]
pats = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(p in n for p in pats):
        d[(n[:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:70s} grid {k[1]:>8s} vgpr {k[2]:>4s} n {len(v):4d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
