#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and occupancy from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) joined with the kernel durations of the
same CSV -> profiles/<tag>_pmc_mfma.{md,json}. Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per
32x32x16 bf16 MFMA, summed over the SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)     share of SIMD-cycles with the matrix pipe busy
  waves_per_simd = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)                 average resident waves per SIMD (occupancy / 8)
  wait_share     = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                      share of wave time parked in s_waitcnt / barriers"""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("out_prefix"); ap.add_argument("--steps", type=int, default=1)
a = ap.parse_args()


def norm(name):
    n = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    return re.sub(r"^void ", "", n)[:90]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(a.csv)):
    k = norm(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    simd_cycles = gui * 1024.0
    rows.append({"kernel": k, "launches": cnt[k], "avg_us": dur[k] / max(cnt[k], 1), "total_us": dur[k],
                 "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles if simd_cycles else None,
                 "waves_per_simd": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles if simd_cycles else None,
                 "wait_share": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                 "mfma_mops_bf16": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / max(cnt[k], 1)})
rows.sort(key=lambda r: -r["total_us"])
json.dump({"kernels": rows}, open(a.out_prefix + ".json", "w"), indent=1)
with open(a.out_prefix + ".md", "w") as o:
    o.write("# MFMA utilisation / occupancy per kernel (rocprofv3 --pmc, one pass; see tools/pmc_mfma_summary.py for the formulas)\n\n")
    o.write("| kernel | launches | avg us (profiled) | MFMA busy | waves / SIMD | wave time waiting |\n|---|---:|---:|---:|---:|---:|\n")
    f = lambda v, p: "-" if v is None else (f"{100 * v:.1f} %" if p else f"{v:.2f}")
    for r in rows[:40]:
        o.write(f"| {r['kernel']} | {r['launches']} | {r['avg_us']:.1f} | {f(r['mfma_util'], 1)} | {f(r['waves_per_simd'], 0)} | {f(r['wait_share'], 1)} |\n")
print(open(a.out_prefix + ".md").read()[:4000])
