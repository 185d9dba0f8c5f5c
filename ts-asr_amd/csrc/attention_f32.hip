// Attention in EXACT fp32 arithmetic (no matrix cores, no bf16 rounding of any operand): the parity mode of RelPosMHAXL and the
// cross-attention speaker injection.
//
// Replaces, for fp32 activations, the same reference code as csrc/attention.hip - SB/nnet/attention.py:586-633 (q+u / q+v, matrix_ac,
// matrix_bd + rel_shift :468-483, scale, -inf masks, softmax, dropout, .V) - and, with pk == NULL and Tq != Tk, the core of
// torch.nn.MultiheadAttention as models/conformer.py:263-266 (`cross_attention` injection) calls it. csrc/attention.hip rounds q+u, q+v,
// k, p, v, the probabilities and dS to bf16 for its MFMA contractions whatever the storage type (5e-2-class agreement with the reference
// on gradients); here every product and sum is fp32 (1e-5-class), at VALU speed: this is the checker-grade path of `compute_dtype: fp32`
// runs, not the benchmarked one.
//
// Layout: generic strided q / k / v (element strides: batch, row, head), so the same kernels take the per-head interleaved qkv tensor
// of RelPosMHAXL and the separate q / kv tensors of the cross-attention. Flash-style: a workgroup owns 16 query rows of one (b, h) and
// walks the keys in tiles of 64 through LDS (K, V, and the 79 rows of the positional band the tile can touch), online softmax per row
// in a wave, nothing of size Tq x Tk in the forward. The backward is three deterministic passes (no float atomics): query-major (dQ,
// bias-gradient partials; writes dropped probabilities and scaled dS to the workspace), key-major (dK, dV), and - with positions - one
// workgroup per 16 rows of d(pk).
#include "common.h"

namespace {

constexpr int QB = 16;       // query rows per workgroup (4 per wave)
constexpr int KT = 64;       // keys per tile = lanes
constexpr int DMAX = 64;     // head dim <= 64 (lane = d)
constexpr int LDD = DMAX + 1;

struct AttnArgs {
    const void *q, *k, *v, *pk;           // pk [2*Tk-1, H*Dh] or NULL
    const float *bu, *bv;                 // [H*Dh] or NULL (zero)
    const int *key_lens;                  // [B] or NULL
    long long bq, bk, bv_;                // batch strides (elements)
    long long ldq, ldk, ldv;              // row strides
    int hq, hk, hv;                       // head strides
    int B, Tq, Tk, H, Dh, causal;
    float scale, pdrop;
    unsigned long long seed;
    const unsigned long long *seed_dev;
};

__device__ __forceinline__ int causal_limit(int i, int causal) { return causal <= 1 ? i : (i / causal + 1) * causal - 1; }

template <typename T>
__device__ __forceinline__ void load_rows(float (*dst)[LDD], const T *src, long long ld, int row0, int nrows, int nvalid, int Dh) {
    // dst[r][d] = src[(row0 + r) * ld + d] for r < nrows (rows >= nvalid read as 0)
    for (int e = threadIdx.x; e < nrows * DMAX; e += 256) {
        const int r = e / DMAX, d = e % DMAX;
        float v = 0.f;
        if (d < Dh && row0 + r < nvalid && row0 + r >= 0) v = ld1(src + (long long)(row0 + r) * ld + d);
        dst[r][d] = v;
    }
}

// scores of query row `ii` of the block against the 64 keys of the tile (lane = key): s = scale * ((q+u).k + (q+v).p_rel), -inf if masked
template <bool POS>
__device__ __forceinline__ float score_row(const float (*Qu)[LDD], const float (*Qv)[LDD], const float (*Ks)[LDD], const float (*Ps)[LDD],
                                           int ii, int lane, int Dh, float scale, bool masked) {
    float s = 0.f;
    for (int d = 0; d < Dh; ++d) {
        s += Qu[ii][d] * Ks[lane][d];
        if (POS) s += Qv[ii][d] * Ps[lane - ii + QB - 1][d];
    }
    return masked ? -INFINITY : s * scale;
}

template <typename T, bool POS>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(AttnArgs a, T *__restrict__ out, float *__restrict__ lse) {
    __shared__ float Ks[KT][LDD], Vs[KT][LDD], Ps[POS ? KT + QB - 1 : 1][LDD], Qu[QB][LDD], Qv[POS ? QB : 1][LDD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * QB;
    const int Dh = a.Dh, Tq = a.Tq, Tk = a.Tk;
    const int klen = a.key_lens ? min(a.key_lens[b], Tk) : Tk;
    const T *q = (const T *)a.q + b * a.bq + (long long)h * a.hq, *k = (const T *)a.k + b * a.bk + (long long)h * a.hk,
            *v = (const T *)a.v + b * a.bv_ + (long long)h * a.hv;
    const T *pk = POS ? (const T *)a.pk + (long long)h * Dh : nullptr;
    for (int e = threadIdx.x; e < QB * DMAX; e += 256) {
        const int r = e / DMAX, d = e % DMAX;
        const float qv = (d < Dh && i0 + r < Tq) ? ld1(q + (long long)(i0 + r) * a.ldq + d) : 0.f;
        Qu[r][d] = qv + ((a.bu && d < Dh) ? a.bu[h * Dh + d] : 0.f);
        if (POS) Qv[r][d] = qv + ((a.bv && d < Dh) ? a.bv[h * Dh + d] : 0.f);
    }
    unsigned long long seed = a.seed;
    if (a.seed_dev) seed += *a.seed_dev;
    const unsigned thr = drop_thr16(a.pdrop);
    const float ks = drop_scale16(thr);
    const DropKey dk = drop_key(seed);
    float m[4], l[4], o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; o[r] = 0.f; }
    for (int j0 = 0; j0 < Tk; j0 += KT) {
        __syncthreads();
        load_rows(Ks, k, a.ldk, j0, KT, Tk, Dh);
        load_rows(Vs, v, a.ldv, j0, KT, Tk, Dh);
        if (POS) load_rows(Ps, pk, (long long)a.H * Dh, j0 - (i0 + QB - 1) + Tk - 1, KT + QB - 1, 2 * Tk - 1, Dh);
        __syncthreads();
        const int j = j0 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = wave * 4 + r, i = i0 + ii;
            if (i >= Tq) continue;                          // wave-uniform
            const bool masked = j >= klen || (a.causal && j > causal_limit(i, a.causal));
            const float s = score_row<POS>(Qu, Qv, Ks, Ps, ii, lane, Dh, a.scale, masked);
            const float mn = fmaxf(m[r], wave_max(s));
            if (mn == -INFINITY) continue;                   // every key so far is masked
            const float alpha = m[r] == -INFINITY ? 0.f : expf(m[r] - mn);
            const float p = masked ? 0.f : expf(s - mn);
            l[r] = l[r] * alpha + wave_sum(p);
            m[r] = mn;
            float pd = p;
            if (thr) pd = drop_keep1(((unsigned long long)(b * a.H + h) * Tq + i) * Tk + j, dk, thr) ? p * ks : 0.f;
            float acc = o[r] * alpha;
            for (int jj = 0; jj < KT; ++jj) acc += lane_bcast(pd, jj) * Vs[jj][lane];
            o[r] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 4 + r;
        if (i >= Tq) continue;
        if (lane < Dh) st1(out + ((long long)b * Tq + i) * a.H * Dh + h * Dh + lane, l[r] > 0.f ? o[r] / l[r] : 0.f);
        if (lane == 0) lse[((long long)b * a.H + h) * Tq + i] = l[r] > 0.f ? m[r] + logf(l[r]) : -INFINITY;
    }
}

// ---- backward, pass A: query-major ------------------------------------------------------------------------------------------------------
// dq rows (complete); PD[b,h,i,j] = dropped probability, DS[b,h,i,j] = scale * dS (both fp32 [B,H,Tq,Tk]); per-workgroup partial sums of
// d(bias_u) / d(bias_v) -> part[(b * nqb + qb)][2][H*Dh] (this workgroup's h slice)
template <typename T, bool POS>
__global__ __launch_bounds__(256) void attn_f32_bwd_q_kernel(AttnArgs a, const T *__restrict__ out, const T *__restrict__ dout,
                                                             const float *__restrict__ lse, T *__restrict__ dq, long long bdq, long long lddq,
                                                             int hdq, float *__restrict__ PD, float *__restrict__ DS, float *__restrict__ part) {
    __shared__ float Ks[KT][LDD], Vs[KT][LDD], Ps[POS ? KT + QB - 1 : 1][LDD], Qu[QB][LDD], Qv[POS ? QB : 1][LDD], dO[QB][LDD];
    __shared__ float red[4][2][DMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * QB;
    const int Dh = a.Dh, Tq = a.Tq, Tk = a.Tk, HD = a.H * Dh;
    const int klen = a.key_lens ? min(a.key_lens[b], Tk) : Tk;
    const T *q = (const T *)a.q + b * a.bq + (long long)h * a.hq, *k = (const T *)a.k + b * a.bk + (long long)h * a.hk,
            *v = (const T *)a.v + b * a.bv_ + (long long)h * a.hv;
    const T *pk = POS ? (const T *)a.pk + (long long)h * Dh : nullptr;
    for (int e = threadIdx.x; e < QB * DMAX; e += 256) {
        const int r = e / DMAX, d = e % DMAX;
        const bool ok = d < Dh && i0 + r < Tq;
        const float qv = ok ? ld1(q + (long long)(i0 + r) * a.ldq + d) : 0.f;
        Qu[r][d] = qv + ((a.bu && d < Dh) ? a.bu[h * Dh + d] : 0.f);
        if (POS) Qv[r][d] = qv + ((a.bv && d < Dh) ? a.bv[h * Dh + d] : 0.f);
        dO[r][d] = ok ? ld1(dout + ((long long)b * Tq + i0 + r) * HD + h * Dh + d) : 0.f;
    }
    unsigned long long seed = a.seed;
    if (a.seed_dev) seed += *a.seed_dev;
    const unsigned thr = drop_thr16(a.pdrop);
    const float ks = drop_scale16(thr);
    const DropKey dk = drop_key(seed);
    float Di[4], Li[4], dqa[4], dua = 0.f, dva = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 4 + r;
        float t = 0.f;
        if (i < Tq && lane < Dh) t = ld1(dout + ((long long)b * Tq + i) * HD + h * Dh + lane) * ld1(out + ((long long)b * Tq + i) * HD + h * Dh + lane);
        Di[r] = wave_sum(t);
        Li[r] = i < Tq ? lse[((long long)b * a.H + h) * Tq + i] : 0.f;
        dqa[r] = 0.f;
    }
    for (int j0 = 0; j0 < Tk; j0 += KT) {
        __syncthreads();
        load_rows(Ks, k, a.ldk, j0, KT, Tk, Dh);
        load_rows(Vs, v, a.ldv, j0, KT, Tk, Dh);
        if (POS) load_rows(Ps, pk, (long long)HD, j0 - (i0 + QB - 1) + Tk - 1, KT + QB - 1, 2 * Tk - 1, Dh);
        __syncthreads();
        const int j = j0 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = wave * 4 + r, i = i0 + ii;
            if (i >= Tq) continue;
            const bool masked = j >= klen || (a.causal && j > causal_limit(i, a.causal));
            const float s = score_row<POS>(Qu, Qv, Ks, Ps, ii, lane, Dh, a.scale, masked);
            const float p = (masked || Li[r] == -INFINITY) ? 0.f : expf(s - Li[r]);
            float dp = 0.f;
            for (int d = 0; d < Dh; ++d) dp += dO[ii][d] * Vs[lane][d];
            float keepf = 1.f;
            if (thr) keepf = drop_keep1(((unsigned long long)(b * a.H + h) * Tq + i) * Tk + j, dk, thr) ? ks : 0.f;
            const float pd = p * keepf;
            const float dsc = p * (dp * keepf - Di[r]) * a.scale;
            if (j < Tk) {
                const long long w = (((long long)b * a.H + h) * Tq + i) * Tk + j;
                PD[w] = pd;
                DS[w] = dsc;
            }
            float acc = dqa[r];
            for (int jj = 0; jj < KT; ++jj) {
                const float g = lane_bcast(dsc, jj);
                const float kk = Ks[jj][lane];
                acc += g * kk;
                dua += g * kk;
                if (POS) {
                    const float pp = Ps[jj - ii + QB - 1][lane];
                    acc += g * pp;
                    dva += g * pp;
                }
            }
            dqa[r] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 4 + r;
        if (i < Tq && lane < Dh) st1(dq + b * bdq + (long long)i * lddq + (long long)h * hdq + lane, dqa[r]);
    }
    red[wave][0][lane] = dua;
    red[wave][1][lane] = dva;
    __syncthreads();
    if (wave < 2 && lane < Dh) {
        const float t = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
        part[(((long long)b * gridDim.x + blockIdx.x) * 2 + wave) * HD + h * Dh + lane] = t;
    }
}

// ---- pass B: key-major: dk_j = sum_i DS[i][j] (q_i + u), dv_j = sum_i PD[i][j] dout_i --------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_f32_bwd_kv_kernel(AttnArgs a, const T *__restrict__ dout, const float *__restrict__ PD,
                                                              const float *__restrict__ DS, T *__restrict__ dk, long long bdk, long long lddk,
                                                              int hdk, T *__restrict__ dv, long long bdv, long long lddv, int hdv) {
    __shared__ float Qs[KT][LDD], dO[KT][LDD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * QB;
    const int Dh = a.Dh, Tq = a.Tq, Tk = a.Tk, HD = a.H * Dh;
    const T *q = (const T *)a.q + b * a.bq + (long long)h * a.hq;
    float dka[4], dva[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dka[r] = 0.f; dva[r] = 0.f; }
    for (int i0 = 0; i0 < Tq; i0 += KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < KT * DMAX; e += 256) {
            const int r = e / DMAX, d = e % DMAX;
            const bool ok = d < Dh && i0 + r < Tq;
            Qs[r][d] = ok ? ld1(q + (long long)(i0 + r) * a.ldq + d) + (a.bu ? a.bu[h * Dh + d] : 0.f) : 0.f;
            dO[r][d] = ok ? ld1(dout + ((long long)b * Tq + i0 + r) * HD + h * Dh + d) : 0.f;
        }
        __syncthreads();
        const int i = i0 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wave * 4 + r;
            if (j >= Tk) continue;
            float ds = 0.f, pd = 0.f;
            if (i < Tq) {
                const long long w = (((long long)b * a.H + h) * Tq + i) * Tk + j;
                ds = DS[w];
                pd = PD[w];
            }
            float ak = dka[r], av = dva[r];
            for (int ii = 0; ii < KT; ++ii) {
                ak += lane_bcast(ds, ii) * Qs[ii][lane];
                av += lane_bcast(pd, ii) * dO[ii][lane];
            }
            dka[r] = ak; dva[r] = av;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + wave * 4 + r;
        if (j < Tk && lane < Dh) {
            st1(dk + b * bdk + (long long)j * lddk + (long long)h * hdk + lane, dka[r]);
            st1(dv + b * bdv + (long long)j * lddv + (long long)h * hdv + lane, dva[r]);
        }
    }
}

// ---- pass C: dpk[r][h*Dh + d] = sum_b sum_i DS[b,h,i, r + i - (T-1)] (q_i + v)[d]  (self-attention: Tq == Tk == T) -------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_f32_bwd_pk_kernel(AttnArgs a, const float *__restrict__ DS, T *__restrict__ dpk) {
    __shared__ float Qs[KT][LDD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, r0 = blockIdx.x * QB;
    const int Dh = a.Dh, Tn = a.Tq, HD = a.H * Dh, R = 2 * Tn - 1;
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const T *q = (const T *)a.q + b * a.bq + (long long)h * a.hq;
        for (int i0 = 0; i0 < Tn; i0 += KT) {
            __syncthreads();
            for (int e = threadIdx.x; e < KT * DMAX; e += 256) {
                const int rr = e / DMAX, d = e % DMAX;
                Qs[rr][d] = (d < Dh && i0 + rr < Tn) ? ld1(q + (long long)(i0 + rr) * a.ldq + d) + (a.bv ? a.bv[h * Dh + d] : 0.f) : 0.f;
            }
            __syncthreads();
            const int i = i0 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rel = r0 + wave * 4 + r;
                if (rel >= R) continue;
                const int j = rel + i - (Tn - 1);
                float ds = 0.f;
                if (i < Tn && j >= 0 && j < Tn) ds = DS[(((long long)b * a.H + h) * Tn + i) * Tn + j];
                float t = acc[r];
                for (int ii = 0; ii < KT; ++ii) t += lane_bcast(ds, ii) * Qs[ii][lane];
                acc[r] = t;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rel = r0 + wave * 4 + r;
        if (rel < R && lane < Dh) st1(dpk + (long long)rel * HD + h * Dh + lane, acc[r]);
    }
}

int check_common(const char *who, const void *q, const void *k, const void *v, int B, int Tq, int Tk, int H, int Dh, int io_dtype, const void *pk) {
    TSASR_CHECK_ARG(q && k && v, "%s: null pointer", who);
    TSASR_CHECK_ARG(B > 0 && Tq > 0 && Tk > 0 && H > 0 && Dh > 0 && Dh <= DMAX, "%s: bad sizes B=%d Tq=%d Tk=%d H=%d Dh=%d (Dh <= %d)", who, B, Tq, Tk, H, Dh, DMAX);
    TSASR_CHECK_ARG(io_dtype == TSASR_F32 || io_dtype == TSASR_BF16, "%s: bad io_dtype", who);
    TSASR_CHECK_ARG(!pk || Tq == Tk, "%s: relative positions need self-attention (Tq == Tk)", who);
    TSASR_CHECK_ARG((long long)B * H * Tq * Tk < (1ll << 40), "%s: problem too large", who);
    return 0;
}

AttnArgs make_args(const void *q, const void *k, const void *v, const long long *strides, const void *pk, const float *bu, const float *bv,
                   const int *key_lens, int B, int Tq, int Tk, int H, int Dh, float scale, int causal, float pdrop, unsigned long long seed,
                   const unsigned long long *seed_dev) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.pk = pk; a.bu = bu; a.bv = bv; a.key_lens = key_lens;
    a.bq = strides[0]; a.ldq = strides[1]; a.hq = (int)strides[2];
    a.bk = strides[3]; a.ldk = strides[4]; a.hk = (int)strides[5];
    a.bv_ = strides[6]; a.ldv = strides[7]; a.hv = (int)strides[8];
    a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.Dh = Dh; a.causal = causal; a.scale = scale; a.pdrop = pdrop; a.seed = seed; a.seed_dev = seed_dev;
    return a;
}

}  // namespace

extern "C" {

/* strides[9] (HOST, elements) = {q batch, q row, q head, k batch, k row, k head, v batch, v row, v head}. */
int tsasr_attn_f32_fwd(const void *q, const void *k, const void *v, const long long *strides, const void *pk, const float *bias_u,
                       const float *bias_v, const int32_t *key_lens, void *out, float *lse, int B, int Tq, int Tk, int H, int Dh, float scale,
                       int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream) {
    if (int rc = check_common("tsasr_attn_f32_fwd", q, k, v, B, Tq, Tk, H, Dh, io_dtype, pk)) return rc;
    TSASR_CHECK_ARG(out && lse && strides, "tsasr_attn_f32_fwd: null pointer");
    const AttnArgs a = make_args(q, k, v, strides, pk, bias_u, bias_v, key_lens, B, Tq, Tk, H, Dh, scale, causal, pdrop, seed, seed_dev);
    dim3 grid(cdiv(Tq, QB), H, B);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32) {
        if (pk) attn_f32_fwd_kernel<float, true><<<grid, 256, 0, st>>>(a, (float *)out, lse);
        else attn_f32_fwd_kernel<float, false><<<grid, 256, 0, st>>>(a, (float *)out, lse);
    } else {
        if (pk) attn_f32_fwd_kernel<bf16_t, true><<<grid, 256, 0, st>>>(a, (bf16_t *)out, lse);
        else attn_f32_fwd_kernel<bf16_t, false><<<grid, 256, 0, st>>>(a, (bf16_t *)out, lse);
    }
    TSASR_CHECK_LAUNCH("tsasr_attn_f32_fwd");
    return 0;
}

size_t tsasr_attn_f32_bwd_workspace_bytes(int B, int Tq, int Tk, int H, int Dh) {
    const size_t plane = align_up((size_t)B * H * Tq * Tk * sizeof(float), 256);
    return 2 * plane + align_up((size_t)B * cdiv(Tq, QB) * 2 * H * Dh * sizeof(float), 256);
}

/* dstrides[9] as strides, for dq / dk / dv (all fully written for rows < Tq / Tk). dpk [2*Tk-1, H*Dh] (io_dtype) when pk != NULL.
 * d_bias_u / d_bias_v fp32 [H*Dh] (NULL = not wanted): written through the deferrable batched reduction (csrc/reduce.hip). */
int tsasr_attn_f32_bwd(const void *q, const void *k, const void *v, const long long *strides, const void *pk, const float *bias_u,
                       const float *bias_v, const int32_t *key_lens, const void *out, const void *dout, const float *lse, void *dq, void *dk,
                       void *dv, const long long *dstrides, void *dpk, float *d_bias_u, float *d_bias_v, int B, int Tq, int Tk, int H, int Dh,
                       float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype,
                       void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_common("tsasr_attn_f32_bwd", q, k, v, B, Tq, Tk, H, Dh, io_dtype, pk)) return rc;
    TSASR_CHECK_ARG(out && dout && lse && dq && dk && dv && strides && dstrides && (!pk || dpk), "tsasr_attn_f32_bwd: null pointer");
    TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_attn_f32_bwd_workspace_bytes(B, Tq, Tk, H, Dh), "tsasr_attn_f32_bwd: workspace too small");
    const AttnArgs a = make_args(q, k, v, strides, pk, bias_u, bias_v, key_lens, B, Tq, Tk, H, Dh, scale, causal, pdrop, seed, seed_dev);
    const size_t plane = align_up((size_t)B * H * Tq * Tk * sizeof(float), 256);
    float *PD = (float *)workspace, *DS = (float *)((char *)workspace + plane), *part = (float *)((char *)workspace + 2 * plane);
    hipStream_t st = (hipStream_t)stream;
    const int nqb = cdiv(Tq, QB), HD = H * Dh;
    dim3 gq(nqb, H, B), gk(cdiv(Tk, QB), H, B), gp(cdiv(2 * Tk - 1, QB), H);
#define TSASR_ATTN_F32_BWD(TT)                                                                                                                   \
    do {                                                                                                                                         \
        if (pk) attn_f32_bwd_q_kernel<TT, true><<<gq, 256, 0, st>>>(a, (const TT *)out, (const TT *)dout, lse, (TT *)dq, dstrides[0], dstrides[1], \
                                                                    (int)dstrides[2], PD, DS, part);                                            \
        else attn_f32_bwd_q_kernel<TT, false><<<gq, 256, 0, st>>>(a, (const TT *)out, (const TT *)dout, lse, (TT *)dq, dstrides[0], dstrides[1],  \
                                                                  (int)dstrides[2], PD, DS, part);                                              \
        attn_f32_bwd_kv_kernel<TT><<<gk, 256, 0, st>>>(a, (const TT *)dout, PD, DS, (TT *)dk, dstrides[3], dstrides[4], (int)dstrides[5],          \
                                                       (TT *)dv, dstrides[6], dstrides[7], (int)dstrides[8]);                                   \
        if (pk) attn_f32_bwd_pk_kernel<TT><<<gp, 256, 0, st>>>(a, DS, (TT *)dpk);                                                                 \
    } while (0)
    if (io_dtype == TSASR_F32) TSASR_ATTN_F32_BWD(float);
    else TSASR_ATTN_F32_BWD(bf16_t);
#undef TSASR_ATTN_F32_BWD
    TSASR_CHECK_LAUNCH("tsasr_attn_f32_bwd");
    // bias gradients: sums over the (b, query block) partial rows, fixed order
    if (d_bias_u) tsasr_reduce_submit(part, d_bias_u, 2ll * HD, B * nqb, HD, 0, st);
    if (d_bias_v) tsasr_reduce_submit(part + HD, d_bias_v, 2ll * HD, B * nqb, HD, 0, st);
    return 0;
}

}  // extern "C"
