for i in 1 2 3 4; do REPS=${REPS:-10} timeout -k 10 300 python tools/scratch/param_det.py 2>&1 | grep "^rep" > gpurun_out/pd_$i.txt; done
python - <<'PY'
def parse(p):
    out=[]
    for l in open(p):
        t=l.split(); out.append((t[1], t[3], dict(x.rsplit(":",1) for x in t[4:])))
    return out
runs=[parse(f"gpurun_out/pd_{i}.txt") for i in (1,2,3,4)]
for rows in zip(*runs):
    ra,la,ga=rows[0]
    msg=[]
    for k,(r,l,g) in enumerate(rows[1:],2):
        d=[n for n in ga if ga[n]!=g[n]]
        msg.append(f"vs{k}: loss_eq={la==l} ndiff={len(d)} {d[:4]}")
    print(ra, " | ".join(msg))
PY
