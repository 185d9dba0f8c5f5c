import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
ops = importlib.import_module("ts-asr_amd.ops")
g = np.load("tests/golden/c1_augment.npz")
key = "sa_recipe_0"
words = [int(g[key+"_c"][0]), int(g[key+"_w"][0])]
for k in ("_flen","_fpos","_tlen","_tpos"): words += g[key+k].reshape(-1).tolist()
print(words)
p = torch.tensor(words, dtype=torch.int32, device="cuda:0")
x = torch.from_numpy(g["sa_x"]).cuda()
y = ops.spec_augment_apply(x, p, 2, 2, False).cpu().numpy()
ref = g[key+"_y"]
d = np.abs(y-ref)
bad = np.argwhere(d > 1e-5)
print(len(bad), "bad; t values:", sorted(set(bad[:,1].tolist()))[:50], "f values", sorted(set(bad[:,2].tolist()))[:90])
print("b", sorted(set(bad[:,0].tolist())))
b,t,f = bad[0]; print(y[b,t,f], ref[b,t,f]); 
vals, cnt = np.unique(ref[tuple(bad.T)], return_counts=True); print(vals[:5], cnt[:5])
vals, cnt = np.unique(y[tuple(bad.T)], return_counts=True); print(vals[:5], cnt[:5])
f32 = np.float32
xs = g["sa_x"]
def taps(t):
    A=f32(-0.75); one=f32(1)
    def near(x): return ((A+f32(2))*x-(A+f32(3)))*x*x+one
    def far(x): return ((A*x-f32(5)*A)*x+f32(8)*A)*x-f32(4)*A
    return far(t+one),near(t),near(one-t),far(f32(2)-t)
c, w, T_ = words[0], words[1], 200
for (b,t,f) in bad[:6].tolist() + bad[-3:].tolist():
    in_len,out_len,d,base = (c,w,t,0) if t < w else (T_-c,T_-w,t-w,c)
    scale=f32(in_len-1)/f32(out_len-1); src=f32(scale*f32(d)); i0=int(np.floor(src)); tt=f32(src-i0); ws=taps(tt)
    idx=[base+min(max(i0-1+k,0),in_len-1) for k in range(4)]
    print((b,t,f), "kernel", y[b,t,f], "ref", ref[b,t,f], "emul", sum(f32(ws[k])*xs[b,idx[k],f] for k in range(4)), "src", src, "taps", idx, [xs[b,i,f] for i in idx])
