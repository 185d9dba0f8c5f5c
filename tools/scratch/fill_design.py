"""Fill DESIGN.md's @D?@ / @V?@ / @NGPU@ placeholders from profiles/r05_bench*.json and the last full GPU test log (scratch)."""
import json, re, sys
s = open("DESIGN.md").read()
for i, c in enumerate(["", "_ragged", "_accum4", "_none", "_pretrained", "_longform", "_longform_chunk40"]):
    d = json.loads(open(f"profiles/r05_bench{c}.json").read().strip().split("\n")[-1])
    s = s.replace(f"@D{i}@", f"{d['ms_per_step']:.2f}").replace(f"@V{i}@", f"{d['value'] / 1e6:.2f}")
if len(sys.argv) > 1:
    s = s.replace("@NGPU@", sys.argv[1])
open("DESIGN.md", "w").write(s)
