import numpy as np
np.seterr(over="ignore")
U = np.uint32
def drop_hash(ctr_lo, ctr_hi, k0, k1):
    x = (ctr_lo + k0).astype(U)
    x ^= x >> U(16); x = (x * U(0x7feb352d)).astype(U)
    x ^= x >> U(15); x = (x + (k1 ^ ctr_hi)).astype(U)
    x = (x * U(0x846ca68b)).astype(U); x ^= x >> U(16)
    return x
def lite1(y):   # one multiply
    y = y.copy(); y ^= y >> U(15); y = (y * U(0x2c1b3c6d)).astype(U); y ^= y >> U(16); return y
def lite1b(y):
    y = y.copy(); y ^= y >> U(16); y = (y * U(0x7feb352d)).astype(U); y ^= y >> U(15); return y
def rot(x, r): return ((x << U(r)) | (x >> U(32 - r))).astype(U)
def arx(a, b, rounds):
    a = a.copy(); b = b.copy()
    R = [13, 15, 26, 6, 17, 29, 16, 24]
    for i in range(rounds):
        a = (a + b).astype(U); b = rot(b, R[i % 8]) ^ a
    return a, b
def stats(mask, name):
    m = mask.astype(np.float64); n = m.size
    def corr(a, b):
        a = a - a.mean(); b = b - b.mean(); return float((a * b).mean() / (a.std() * b.std()))
    lim = 6 / n ** 0.5
    vals = dict(mean=m.mean(), c_col=corr(m[:, :-1], m[:, 1:]), c_col2=corr(m[:, :-2], m[:, 2:]), c_col4=corr(m[:, :-4], m[:, 4:]), c_col8=corr(m[:, :-8], m[:, 8:]),
                c_col16=corr(m[:, :-16], m[:, 16:]), c_col32=corr(m[:, :-32], m[:, 32:]), c_row=corr(m[:-1], m[1:]), c_row2=corr(m[:-2], m[2:]), c_diag=corr(m[:-1, :-1], m[1:, 1:]),
                c_adiag=corr(m[:-1, 1:], m[1:, :-1]))
    bad = [k for k, v in vals.items() if k != "mean" and abs(v) > lim]
    print(f"{name:28s} mean {vals['mean']:.5f} lim {lim:.2e} " + " ".join(f"{k}={v:+.1e}" for k, v in vals.items() if k != "mean") + ("  BAD " + ",".join(bad) if bad else "  ok"))
rows, T = 4096, 1024          # rows = (b,h,i) flattened
thr = U(6554)
k0, k1 = U(0x1234abcd), U(0x9e3779b1)
rid = np.arange(rows, dtype=np.uint64)
# reference: current scheme, hash per pair of consecutive keys
idx = (rid[:, None] * np.uint64(T) + np.arange(T, dtype=np.uint64)[None, :])
c = idx >> np.uint64(1)
h = drop_hash((c & np.uint64(0xffffffff)).astype(U), (c >> np.uint64(32)).astype(U), k0, k1)
u16 = np.where((idx & np.uint64(1)) == 1, h >> U(16), h & U(0xffff))
stats(u16 >= thr, "current drop_hash")
# candidates: row state s = drop_hash(rid), word k of the row: w = mix(s + k*GOLD); elements 2k, 2k+1 <- halves
s = drop_hash((rid & np.uint64(0xffffffff)).astype(U), (rid >> np.uint64(32)).astype(U), k0, k1)
s2 = drop_hash(((rid + np.uint64(1 << 40)) & np.uint64(0xffffffff)).astype(U), ((rid + np.uint64(1 << 40)) >> np.uint64(32)).astype(U), k0, k1)
kk = np.arange(T // 2, dtype=U)
for nm, fn in (("lite1 weyl", lite1), ("lite1b weyl", lite1b)):
    w = fn((s[:, None] + kk[None, :] * U(0x9E3779B9)).astype(U))
    u = np.empty((rows, T), dtype=U); u[:, 0::2] = w & U(0xffff); u[:, 1::2] = w >> U(16)
    stats(u >= thr, nm)
for r in (2, 3, 4, 5, 6):
    a, b = arx((s[:, None] + kk[None, ::2] * U(0x9E3779B9)).astype(U), (s2[:, None] ^ (kk[None, ::2] * U(0x85EBCA6B))).astype(U), r)
    u = np.empty((rows, T), dtype=U); u[:, 0::4] = a & U(0xffff); u[:, 1::4] = a >> U(16); u[:, 2::4] = b & U(0xffff); u[:, 3::4] = b >> U(16)
    stats(u >= thr, f"arx {r} rounds")
def mul24(a, b): return ((a & U(0xffffff)).astype(np.uint64) * np.uint64(b & 0xffffff) & np.uint64(0xffffffff)).astype(U)
def lite24(y, K=0xc1b3c6d):
    y = y.copy(); y ^= y >> U(15); y = mul24(y, K); y ^= y >> U(16); return y
def lite24b(y, K=0xc1b3c6d):   # fold the top byte in first
    y = y.copy(); y ^= y >> U(11); y = mul24(y, K); y ^= y >> U(15); return y
def lite24c(y):   # two 24-bit multiplies (both full rate)
    y = y.copy(); y ^= y >> U(15); y = mul24(y, 0xc1b3c6d); y ^= y >> U(13); y = mul24(y, 0x97a2d39); y ^= y >> U(16); return y
for nm, fn in (("lite24", lite24), ("lite24b", lite24b), ("lite24c", lite24c)):
    for seedk in (0, 1):
        ss = s if seedk == 0 else s2
        w = fn((ss[:, None] + kk[None, :] * U(0x9E3779B9)).astype(U))
        u = np.empty((rows, T), dtype=U); u[:, 0::2] = w & U(0xffff); u[:, 1::2] = w >> U(16)
        stats(u >= thr, nm + f" seed{seedk}")
        stats(u >= U(32768), nm + f" seed{seedk} p=.5")
