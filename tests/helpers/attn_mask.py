"""numpy twin of the attention dropout stream (csrc/attn_common.h: attn_row_state / attn_drop_word / attn_pair_mask, on csrc/common.h's
drop_key / drop_hash): keep[b, h, i, j] of a relative-position attention call with call seed `seed` (the value the kernel ends up with:
the seed argument plus the device step counter). Test infrastructure only."""
import numpy as np

U32, U64 = np.uint32, np.uint64
M64 = (1 << 64) - 1


def drop_key(seed):
    z = (int(seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return U32(z & 0xffffffff), U32(z >> 32)


def drop_hash(ctr, k0, k1):
    ctr = np.asarray(ctr, dtype=U64)
    with np.errstate(over="ignore"):
        x = ((ctr & U64(0xffffffff)).astype(U32) + k0).astype(U32)
        x ^= x >> U32(16)
        x = (x * U32(0x7feb352d)).astype(U32)
        x ^= x >> U32(15)
        x = (x + (k1 ^ (ctr >> U64(32)).astype(U32))).astype(U32)
        x = (x * U32(0x846ca68b)).astype(U32)
        x ^= x >> U32(16)
    return x


def thr16(p):
    return min(65535, int(p * 65536.0 + 0.5)) if p > 0 else 0


def keep_mask(B, H, T, p, seed):
    """bool [B, H, T, T]: True where the score element survives dropout."""
    k0, k1 = drop_key(seed)
    rows = np.arange(B * H * T, dtype=U64)
    state = drop_hash(rows, k0, k1)                                   # S(row)
    j = np.arange(T)
    jblk, jl = j >> 5, j & 31
    hh, q, e = (jl >> 2) & 1, jl >> 3, jl & 3                          # key = 32 jblk + 4 hh + 8 q + e
    w = (((jblk * 2 + hh) * 8) + 2 * q + (e >> 1)).astype(U32)
    with np.errstate(over="ignore"):
        y = (state[:, None] + w[None, :] * U32(0x9E3779B9)).astype(U32)
        y ^= y >> U32(15)
        y = ((y & U32(0xffffff)).astype(U64) * U64(0x1b3c6d) & U64(0xffffffff)).astype(U32)
        y ^= y >> U32(16)
    half = np.where((e & 1)[None, :] == 1, y >> U32(16), y & U32(0xffff)).astype(np.uint16).view(np.int16)
    keep = half.astype(np.int32) >= thr16(p) - 32768
    return keep.reshape(B, H, T, T)
