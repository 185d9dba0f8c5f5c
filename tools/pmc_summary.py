#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<tag>_pmc_traffic.{md,json}.
gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md: both counters count KiB; FETCH_SIZE under-reports by 2x."""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser()
ap.add_argument("fetch_csv"); ap.add_argument("write_csv"); ap.add_argument("out_prefix")
a = ap.parse_args()

def norm(name):
    n = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", n)
    if n.startswith("_Z") and m:
        n = m.group(1)
    return n   # bench.py labels GEMM launches by the exact kernel (ring / wave-K / register-staged main loop)

def load(path, scale):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[norm(r["Kernel_Name"])].append(float(r["Counter_Value"]) * scale)
    return agg

f = load(a.fetch_csv, 1024.0 * 2.0)
w = load(a.write_csv, 1024.0)
rows = []
for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
    nf, nw = len(f.get(k, [])), len(w.get(k, []))
    rows.append({"kernel": k, "launches": max(nf, nw), "fetch_bytes_per_launch": sum(f.get(k, [0])) / max(nf, 1),
                 "write_bytes_per_launch": sum(w.get(k, [0])) / max(nw, 1)})
json.dump({"unit": "bytes per launch (FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024)", "kernels": rows}, open(a.out_prefix + ".json", "w"), indent=1)
with open(a.out_prefix + ".md", "w") as o:
    o.write("# HBM-side traffic per kernel launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; gfx950 corrections applied)\n\n")
    o.write("| kernel | launches | fetch MB/launch | write MB/launch |\n|---|---:|---:|---:|\n")
    for r in rows[:60]:
        o.write(f"| {r['kernel'][:80]} | {r['launches']} | {r['fetch_bytes_per_launch']/1e6:.2f} | {r['write_bytes_per_launch']/1e6:.2f} |\n")
print(open(a.out_prefix + ".md").read()[:3500])
