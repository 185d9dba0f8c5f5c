for i in 1 2 3; do TSASR_OVERLAP=${OV:-1} TSASR_EARLY_FLUSH=${EF:-0} REPS=6 timeout -k 10 300 python tools/scratch/grad_det.py 2>&1 | grep "^rep\|graphs\|^B1" > gpurun_out/gdx_$i.txt; done
python - <<'PY'
def parse(p):
    out=[]
    for l in open(p):
        if l.startswith("rep"):
            t=l.split()
            out.append((t[1], t[3], dict(x.rsplit(":",1) for x in t[4:])))
    return out
a,b,c=(parse(f"gpurun_out/gdx_{i}.txt") for i in (1,2,3))
for (ra,la,ga),(rb,lb,gb),(rc,lc,gc) in zip(a,b,c):
    print(" ", ra, la==lb==lc, "| diff12:", [k for k in ga if ga[k]!=gb[k]][:8], "| diff13:", [k for k in ga if ga[k]!=gc[k]][:8])
PY
