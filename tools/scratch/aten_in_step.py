"""List the non-library kernels (at::, rocclr) of ONE replayed step from a rocprofv3 kernel-trace CSV with their time offsets and neighbours (scratch)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))), key=lambda x: x[0])
# steps are separated by the optimizer kernel
ends = [i for i, r in enumerate(rows) if "clip_adamw" in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = step[0][0]
print("kernels in the step:", len(step), " span %.2f ms" % ((step[-1][1] - t0) / 1e6))
for i, (s, e, n, q) in enumerate(step):
    if n.startswith("at::") or "rocclr" in n or "at::native" in n:
        prev = step[i - 1][2][:40] if i else ""
        nxt = step[i + 1][2][:40] if i + 1 < len(step) else ""
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:6.1f} us  q{q}  {n[:70]:70s} | after {prev} | before {nxt}")
