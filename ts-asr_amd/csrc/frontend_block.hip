// One ConvBlock of the convolutional front-end in one pass per direction (gfx950).
//
// Replaces, for speechbrain/lobes/models/convolution.py:187-266 (ConvBlock: convs = Conv2d 3x3 stride 2 -> LayerNorm([F',C]) ->
// LeakyReLU -> Dropout; reduce_conv = Conv2d 1x1 stride 2 -> LayerNorm([F',C]); out = Dropout(convs(x) + reduce_conv(x))), the chain
// conv kernel -> 2 LayerNorm kernels -> dropout-add kernel (7 passes over [B,T',F',C] tensors of 164 MB each at the benchmark's
// first block, 6 more in the backward) with
//   forward : out  = Drop_p2( LN_r(r) + Drop_p1( LeakyReLU( LN_y(y) ) ) )            one write of `out`, nothing else
//   backward: dout -> (dy, dr) and all parameter gradients                            one read of `dout`
// where (y, r) are the two convolution outputs:
//   CONV = true  (block 1, C_in = 1): computed on the fly from the [B,T,F] features (9 + 1 taps per output, filters in registers);
//                the backward recomputes them the same way (bit-identical) and folds (dy, dr) straight into the filter gradients:
//                neither y, r, their normalised forms nor their gradients ever exist in memory.
//   CONV = false (wider blocks): read from the GEMM outputs y1, y2; the backward writes dy1, dy2 for the GEMM backward.
//
// Work split: a row = one (b, t') = F' x C values per branch. 512 threads = 8 waves per workgroup, persistent over rows
// row = blockIdx.x + k * gridDim.x. Wave w owns the frequency positions f' = w, w + 8, ... (NP of them), lane l the channels 2l, 2l+1
// (C = 128), so a wave-instruction stores 256 contiguous bytes, the taps of a position are wave-uniform (one lane loads each tap, the
// values are broadcast through v_readlane into SGPRs), and LayerNorm's gamma/beta [F',C] as well as every per-element gradient
// accumulator live in registers for the whole kernel (each (f', c) belongs to exactly one thread). Statistics are two-pass (mean, then
// centred squares) over register values; two workgroup reductions per row forward, one backward.
#include "common.h"

namespace {

constexpr int FB_THREADS = 512, FB_WAVES = 8, FB_C = 128, FB_MAXNP = 5;

struct FeBlockArgs {
    const void *x;                       // CONV: features [B, Tn, F]
    const void *y1, *y2;                 // !CONV: conv outputs [R, Fo, C]
    const float *w1, *b1, *w2, *b2;      // CONV: [C][3(kf)][3(kt)], [C], [C], [C]
    const float *g1, *be1, *g2, *be2;    // LayerNorm affine [Fo*C] of the two branches
    void *out;                           // forward: [R, Fo, C]
    float *stats;                        // [R][4] = mean_y, rstd_y, mean_r, rstd_r (written forward, read backward)
    const void *dout;                    // backward: [R, Fo, C]
    void *dy1, *dy2;                     // backward, !CONV: [R, Fo, C]
    float *slab;                         // backward: [gridDim.x][slab_width] partial parameter gradients
    long long R;                         // rows = B * To
    int Tn, F, To, Fo, tmode, fmode;
    float slope, eps, p1, p2;
    unsigned long long seed1, seed2;
    const unsigned long long *seed_dev;
};

__device__ __forceinline__ int fb_src_index(int o, int k, int n, int mode) {   // csrc/frontend.hip src_index
    if (mode == 1) { const int i = 2 * o + k - 2; return i < 0 ? -1 : i; }
    int i = 2 * o + k - 1;
    if (mode == 0) { if (i < 0) i = -i; if (i >= n) i = 2 * (n - 1) - i; return i; }
    return (i < 0 || i >= n) ? -1 : i;
}

template <typename T> __device__ __forceinline__ void ld2(const T *p, float (&o)[2]);
template <> __device__ __forceinline__ void ld2<float>(const float *p, float (&o)[2]) { const float2 v = *reinterpret_cast<const float2 *>(p); o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t *p, float (&o)[2]) {
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    o[0] = __uint_as_float(u << 16); o[1] = __uint_as_float(u & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st2(T *p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t *p, float a, float b) {
    bf16x2 o;
    o[0] = (bf16_t)a; o[1] = (bf16_t)b;
    *reinterpret_cast<bf16x2 *>(p) = o;
}

// taps of a whole row in ONE load: lane l = i * 10 + k holds tap k (index kf*3 + kt as the filter is stored; k = 9: the centre sample of
// the 1x1 branch) of this wave's i-th position; v_readlane hands a value to the whole wave where it is used
template <typename T, int NP>
__device__ __forceinline__ float fb_row_taps(const T *__restrict__ x, int b, int to, int wave, int Fo, int Tn, int F, int tmode, int fmode, int lane) {
    const int li = lane < 10 * NP ? lane : 10 * NP - 1;
    const int i = li / 10, k = li - i * 10;
    const int fo = min(wave + FB_WAVES * i, Fo - 1);
    const int kf = k / 3, kt = k - kf * 3;
    const int ti = k < 9 ? fb_src_index(to, kt, Tn, tmode) : 2 * to;
    const int fi = k < 9 ? fb_src_index(fo, kf, F, fmode) : 2 * fo;
    const float raw = ld1(x + ((size_t)b * Tn + max(ti, 0)) * F + max(fi, 0));
    return (ti >= 0 && fi >= 0) ? raw : 0.f;
}

// sum over the workgroup of NV values per thread: wave_sum, one LDS slot per wave, fixed-order sum (every thread gets the totals)
template <int NV>
__device__ __forceinline__ void fb_wg_sum(float (&v)[NV], float *red /*[FB_WAVES][NV]*/, int wave, int lane) {
    if constexpr (NV == 4) wave_sum4(v);
    else {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = wave_sum(v[q]);
    }
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * NV + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FB_WAVES; ++w) s += red[w * NV + q];
        v[q] = s;
    }
}

// keep-factors (ks or 0) of the two channels of a lane from one hash of the pair counter (common.h drop_keep_mask's bit assignment)
__device__ __forceinline__ void fb_keep2(unsigned ctr_lo, unsigned ctr_hi, DropKey k, unsigned thr, float ks, float (&m)[2]) {
    const unsigned h = drop_hash(((unsigned long long)ctr_hi << 32) | ctr_lo, k);
    m[0] = (h & 0xffffu) >= thr ? ks : 0.f;
    m[1] = (h >> 16) >= thr ? ks : 0.f;
}

// FULL: Fo == 8 * NP, every (wave, i) is a real position (no masking anywhere)
template <typename T, bool CONV, int NP, bool FULL>
__global__ __launch_bounds__(FB_THREADS) void fe_block_fwd_kernel(const FeBlockArgs A) {
    __shared__ float red[2][FB_WAVES * 2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c0 = 2 * lane;
    const int Fo = A.Fo, N = Fo * FB_C;
    const float inv_n = 1.f / (float)N;
    // this thread's positions (clamped: an out-of-range position computes on the last one and is masked), affine parameters, filters
    int off[NP];                         // element offset of (fo, c0) inside a row
    bool ok[NP];
    float g1[NP][2], be1[NP][2], g2[NP][2], be2[NP][2];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = wave + FB_WAVES * i;
        ok[i] = FULL || f < Fo;
        off[i] = (ok[i] ? f : Fo - 1) * FB_C + c0;
        ld2(A.g1 + off[i], g1[i]);
        ld2(A.be1 + off[i], be1[i]);
        ld2(A.g2 + off[i], g2[i]);
        ld2(A.be2 + off[i], be2[i]);
    }
    float wf[2][9], bb1[2], ww2[2], bb2[2];
    if (CONV) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wf[j][k] = A.w1[(c0 + j) * 9 + k];
            bb1[j] = A.b1[c0 + j]; ww2[j] = A.w2[c0 + j]; bb2[j] = A.b2[c0 + j];
        }
    }
    const unsigned long long sd = A.seed_dev ? *A.seed_dev : 0ull;
    const unsigned thr1 = drop_thr16(A.p1), thr2 = drop_thr16(A.p2);
    const DropKey k1 = drop_key(A.seed1 + sd), k2 = drop_key(A.seed2 + sd);
    const float ks1 = drop_scale16(thr1), ks2 = drop_scale16(thr2);
    const T *x = (const T *)A.x;
    int b = (int)(blockIdx.x / (unsigned)A.To), to = (int)(blockIdx.x % (unsigned)A.To);
    const int stride_b = (int)(gridDim.x / (unsigned)A.To), stride_t = (int)(gridDim.x % (unsigned)A.To);
    for (long long row = blockIdx.x; row < A.R; row += gridDim.x) {
        const T *y1r = (const T *)A.y1 + row * N, *y2r = (const T *)A.y2 + row * N;
        T *outr = (T *)A.out + row * N;
        float y[NP][2], r[NP][2];
        if (CONV) {
            const float tv = fb_row_taps<T, NP>(x, b, to, wave, Fo, A.Tn, A.F, A.tmode, A.fmode, lane);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float a[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) a[q] = lane_bcast(tv, i * 10 + q);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s = bb1[j];
#pragma unroll
                    for (int k = 0; k < 9; ++k) s += wf[j][k] * a[k];
                    y[i][j] = s;
                    r[i][j] = bb2[j] + ww2[j] * a[9];
                }
            }
            to += stride_t; b += stride_b;
            if (to >= A.To) { to -= A.To; ++b; }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                ld2(y1r + off[i], y[i]);
                ld2(y2r + off[i], r[i]);
            }
        }
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[0] += ok[i] ? y[i][j] : 0.f; s[1] += ok[i] ? r[i][j] : 0.f; }
        fb_wg_sum<2>(s, red[0], wave, lane);
        const float mu_y = s[0] * inv_n, mu_r = s[1] * inv_n;
        float q[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                y[i][j] -= mu_y; r[i][j] -= mu_r;
                q[0] += ok[i] ? y[i][j] * y[i][j] : 0.f;
                q[1] += ok[i] ? r[i][j] * r[i][j] : 0.f;
            }
        fb_wg_sum<2>(q, red[1], wave, lane);
        const float rs_y = rsqrtf(q[0] * inv_n + A.eps), rs_r = rsqrtf(q[1] * inv_n + A.eps);
        if (threadIdx.x == 0) *reinterpret_cast<float4 *>(A.stats + row * 4) = make_float4(mu_y, rs_y, mu_r, rs_r);
        const unsigned long long ctr0 = (unsigned long long)row * (N / 2);    // pair counter of the row's first element
        const unsigned c_lo = (unsigned)ctr0, c_hi = (unsigned)(ctr0 >> 32);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (!ok[i]) continue;                               // wave-uniform
            float m1[2] = {1.f, 1.f}, m2[2] = {1.f, 1.f};
            const unsigned lo = c_lo + (unsigned)(off[i] >> 1), hi = c_hi + (lo < c_lo ? 1u : 0u);
            if (thr1) fb_keep2(lo, hi, k1, thr1, ks1, m1);
            if (thr2) fb_keep2(lo, hi, k2, thr2, ks2, m2);
            float o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float z = y[i][j] * (rs_y * g1[i][j]) + be1[i][j];
                const float act = fmaxf(z, z * A.slope);        // slope < 1 (checked on the host)
                const float rn = r[i][j] * (rs_r * g2[i][j]) + be2[i][j];
                o[j] = (rn + act * m1[j]) * m2[j];
            }
            st2(outr + off[i], o[0], o[1]);
        }
    }
}

// slab row of a workgroup: CONV: [dw1 C*9 | db1 C | dw2 C | db2 C] then [dg1 | dbe1 | dg2 | dbe2] (Fo*C each)
template <typename T, bool CONV, int NP, bool FULL>
__global__ __launch_bounds__(FB_THREADS) void fe_block_bwd_kernel(const FeBlockArgs A) {
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];   // [2][FB_WAVES*4] reductions, then (CONV) [FB_WAVES][C*12] filter-gradient fold
    float *red = fb_lds;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c0 = 2 * lane;
    const int Fo = A.Fo, N = Fo * FB_C;
    const float inv_n = 1.f / (float)N;
    int off[NP];
    bool ok[NP];
    float g1[NP][2], be1[NP][2], g2[NP][2];
    float ag1[NP][2], abe1[NP][2], ag2[NP][2], abe2[NP][2];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = wave + FB_WAVES * i;
        ok[i] = FULL || f < Fo;
        off[i] = (ok[i] ? f : Fo - 1) * FB_C + c0;
        ld2(A.g1 + off[i], g1[i]);
        ld2(A.be1 + off[i], be1[i]);
        ld2(A.g2 + off[i], g2[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) ag1[i][j] = abe1[i][j] = ag2[i][j] = abe2[i][j] = 0.f;
    }
    float wf[2][9], bb1[2], ww2[2], bb2[2];
    float dw[2][9], db1[2], dw2[2], db2[2];
    if (CONV) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) { wf[j][k] = A.w1[(c0 + j) * 9 + k]; dw[j][k] = 0.f; }
            bb1[j] = A.b1[c0 + j]; ww2[j] = A.w2[c0 + j]; bb2[j] = A.b2[c0 + j];
            db1[j] = dw2[j] = db2[j] = 0.f;
        }
    }
    const unsigned long long sd = A.seed_dev ? *A.seed_dev : 0ull;
    const unsigned thr1 = drop_thr16(A.p1), thr2 = drop_thr16(A.p2);
    const DropKey k1 = drop_key(A.seed1 + sd), k2 = drop_key(A.seed2 + sd);
    const float ks1 = drop_scale16(thr1), ks2 = drop_scale16(thr2);
    const T *x = (const T *)A.x;
    int b = (int)(blockIdx.x / (unsigned)A.To), to = (int)(blockIdx.x % (unsigned)A.To);
    const int stride_b = (int)(gridDim.x / (unsigned)A.To), stride_t = (int)(gridDim.x % (unsigned)A.To);
    int par = 0;
    for (long long row = blockIdx.x; row < A.R; row += gridDim.x, par ^= 1) {
        const T *y1r = (const T *)A.y1 + row * N, *y2r = (const T *)A.y2 + row * N, *dor = (const T *)A.dout + row * N;
        T *dy1r = (T *)A.dy1 + row * N, *dy2r = (T *)A.dy2 + row * N;
        const float4 st = *reinterpret_cast<const float4 *>(A.stats + row * 4);
        const float mu_y = st.x, rs_y = st.y, mu_r = st.z, rs_r = st.w;
        float d[NP][2];
#pragma unroll
        for (int i = 0; i < NP; ++i) ld2(dor + off[i], d[i]);   // clamped position: always in range
        float tv = 0.f;
        if (CONV) {
            tv = fb_row_taps<T, NP>(x, b, to, wave, Fo, A.Tn, A.F, A.tmode, A.fmode, lane);
            to += stride_t; b += stride_b;
            if (to >= A.To) { to -= A.To; ++b; }
        }
        const unsigned long long ctr0 = (unsigned long long)row * (N / 2);
        const unsigned c_lo = (unsigned)ctr0, c_hi = (unsigned)(ctr0 >> 32);
        float yh[NP][2], rh[NP][2], gy[NP][2], gr[NP][2];
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float y[2], r[2];
            if (CONV) {
                float a[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) a[q] = lane_bcast(tv, i * 10 + q);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float sacc = bb1[j];
#pragma unroll
                    for (int k = 0; k < 9; ++k) sacc += wf[j][k] * a[k];
                    y[j] = sacc;
                    r[j] = bb2[j] + ww2[j] * a[9];
                }
            } else {
                ld2(y1r + off[i], y);
                ld2(y2r + off[i], r);
            }
            float m1[2] = {1.f, 1.f}, m2[2] = {1.f, 1.f};
            const unsigned lo = c_lo + (unsigned)(off[i] >> 1), hi = c_hi + (lo < c_lo ? 1u : 0u);
            if (thr1) fb_keep2(lo, hi, k1, thr1, ks1, m1);
            if (thr2) fb_keep2(lo, hi, k2, thr2, ks2, m2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float ds = ok[i] ? d[i][j] * m2[j] : 0.f;               // gradient of r_n + Drop(act)
                const float h = (y[j] - mu_y) * rs_y;
                const float z = h * g1[i][j] + be1[i][j];
                const float da = ds * m1[j];
                const float dz = z > 0.f ? da : da * A.slope;
                yh[i][j] = h;
                ag1[i][j] += dz * h;
                abe1[i][j] += dz;
                const float gyv = dz * g1[i][j];
                gy[i][j] = gyv;
                s[0] += gyv; s[1] += gyv * h;
                const float hr = (r[j] - mu_r) * rs_r;
                rh[i][j] = hr;
                ag2[i][j] += ds * hr;
                abe2[i][j] += ds;
                const float grv = ds * g2[i][j];
                gr[i][j] = grv;
                s[2] += grv; s[3] += grv * hr;
            }
        }
        fb_wg_sum<4>(s, red + par * FB_WAVES * 4, wave, lane);
        const float m1y = s[0] * inv_n, m2y = s[1] * inv_n, m1r = s[2] * inv_n, m2r = s[3] * inv_n;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (!ok[i]) continue;                               // wave-uniform
            float dyv[2], drv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dyv[j] = rs_y * (gy[i][j] - m1y - yh[i][j] * m2y);
                drv[j] = rs_r * (gr[i][j] - m1r - rh[i][j] * m2r);
            }
            if (CONV) {
                float a[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) a[q] = lane_bcast(tv, i * 10 + q);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) dw[j][k] += dyv[j] * a[k];
                    db1[j] += dyv[j];
                    dw2[j] += drv[j] * a[9];
                    db2[j] += drv[j];
                }
            } else {
                st2(dy1r + off[i], dyv[0], dyv[1]);
                st2(dy2r + off[i], drv[0], drv[1]);
            }
        }
    }
    // partial parameter gradients of this workgroup
    const int conv_w = CONV ? FB_C * 12 : 0;
    float *mine = A.slab + (size_t)blockIdx.x * (conv_w + 4 * N);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (!ok[i]) continue;
        const int e = off[i];
        *reinterpret_cast<float2 *>(mine + conv_w + e) = make_float2(ag1[i][0], ag1[i][1]);
        *reinterpret_cast<float2 *>(mine + conv_w + N + e) = make_float2(abe1[i][0], abe1[i][1]);
        *reinterpret_cast<float2 *>(mine + conv_w + 2 * N + e) = make_float2(ag2[i][0], ag2[i][1]);
        *reinterpret_cast<float2 *>(mine + conv_w + 3 * N + e) = make_float2(abe2[i][0], abe2[i][1]);
    }
    if (CONV) {   // every wave holds a partial of every channel's 12 filter gradients: fold the 8 waves through LDS in fixed order
        float *fold = fb_lds + 2 * FB_WAVES * 4 + wave * (FB_C * 12);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) fold[(c0 + j) * 9 + k] = dw[j][k];
            fold[FB_C * 9 + c0 + j] = db1[j];
            fold[FB_C * 10 + c0 + j] = dw2[j];
            fold[FB_C * 11 + c0 + j] = db2[j];
        }
        __syncthreads();
        const float *f0 = fb_lds + 2 * FB_WAVES * 4;
        for (int i = threadIdx.x; i < FB_C * 12; i += FB_THREADS) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < FB_WAVES; ++w) acc += f0[w * (FB_C * 12) + i];
            mine[i] = acc;
        }
    }
}

// =====================================================================================================================
// Block 1 (C_in = 1) in bf16 with the two convolutions on the matrix cores (round 5). The kernels above spend most of their vector
// instructions on the 9 + 1 taps (per wave and row: 108 fused multiply-adds + 66 v_readlane forward; 256 + 130 backward, where the filter
// gradients are the same products again) and on drop_hash's quarter-rate multiplies: 731 / 1010 instructions per wave and row, at one
// wave-instruction per CU and cycle = the kernels' whole duration (profiles/r05_notes.md). Here
//   * the taps of a row are staged once in LDS, as P[position][slot] and T[slot][position] (slot k = kf*3 + kt as the filter is stored,
//     slot 9 = 1 at real positions: the bias), by 432 threads that own one (slot, position) each for the whole kernel;
//   * y and r are three v_mfma_f32_16x16x32_bf16 each per wave and row: K = 32 holds the 10 slots twice, against the bf16 high and low
//     parts of the fp32 filters (w = hi + lo to 2^-17), so the result is the fp32 convolution of the bf16 features to ~1e-5;
//   * wave w owns channels 16w .. 16w+15 at all positions. Forward: D^T = W^T . P^T, a lane holds 4 consecutive channels of one position
//     (8-byte stores, float4 parameters). Backward: D = P . W, a lane holds one channel at 4 consecutive positions - exactly the B
//     operand of the filter-gradient MFMA D[slot][c] += sum_p T[slot][p] . dy[p][c] (positions on K), whose accumulators stay in
//     registers for the whole kernel: no cross-wave fold, no tap broadcasts;
//   * LayerNorm statistics in one pass (sum and sum of squares of the fp32 accumulators; padded positions are exact zeros), one workgroup
//     reduction and one barrier per row, which also publishes the next row's taps (double-buffered);
//   * dropout words from the per-row stream of csrc/attn_common.h's form (one strong hash per row on the scalar unit, one full-rate 24-bit
//     multiply per pair of channels): word(pair) = mix24(drop_hash(row, key) + pair_index * 0x9E3779B9).
// =====================================================================================================================
constexpr int F1_PP = 48;                        // padded positions: up to 3 blocks of 16
constexpr int F1_PBYTES = F1_PP * 16 * 2;        // P[48][16] bf16
constexpr int F1_TBYTES = 16 * F1_PP * 2;        // T[16][48] bf16
constexpr int F1_LDS = 21504;      // tiles 2 * (1536 + 1536) + reductions 256; the forward first stages its filters ([2][16][132] floats) and their Gram sums there

__device__ __forceinline__ unsigned fe1_word(unsigned row_key, unsigned cidx) {
    unsigned y = row_key + cidx;
    y ^= y >> 15;
    y = __umul24(y, 0x1b3c6du);
    y ^= y >> 16;
    return y;
}
// filters of this lane as MFMA operand: lane (channel 16*wave + (lane & 15), k-group g = lane >> 4) holds slots 8(g&1) .. +7; g < 2 the bf16
// high parts, g >= 2 the low parts
__device__ __forceinline__ void fe1_filters(const FeBlockArgs &A, int wave, int lane, int centre, bf16x8 &wy, bf16x8 &wr) {
    const int c = 16 * wave + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int slot = 8 * (g & 1) + e;
        const float v = slot < 9 ? A.w1[c * 9 + slot] : (slot == 9 ? A.b1[c] : 0.f);
        const float vr = slot == centre ? A.w2[c] : (slot == 9 ? A.b2[c] : 0.f);
        const bf16_t vh = (bf16_t)v, vrh = (bf16_t)vr;
        wy[e] = g < 2 ? vh : (bf16_t)(v - (float)vh);
        wr[e] = g < 2 ? vrh : (bf16_t)(vr - (float)vrh);
    }
}
// the staging threads' life: thread tid < 9 * 48 owns (slot k, position p); value of row (b, to) = the padded source sample (raw bf16 bits)
struct Fe1Stage {
    int k, p, fi, kt;
    bool on;
    __device__ __forceinline__ void init(const FeBlockArgs &A) {
        const int t = threadIdx.x;
        on = t < 9 * F1_PP;
        k = on ? t / F1_PP : 0;
        p = t - k * F1_PP;
        const int kf = k / 3;
        kt = k - kf * 3;
        fi = (on && p < A.Fo) ? fb_src_index(p, kf, A.F, A.fmode) : -1;
    }
    __device__ __forceinline__ unsigned short load(const FeBlockArgs &A, int b, int to) const {
        const int ti = fb_src_index(to, kt, A.Tn, A.tmode);
        const unsigned short raw = *(reinterpret_cast<const unsigned short *>(A.x) + ((size_t)b * A.Tn + max(ti, 0)) * A.F + max(fi, 0));   // always issued
        return (ti >= 0 && fi >= 0) ? raw : (unsigned short)0;
    }
    __device__ __forceinline__ void store(char *pbuf, char *tbuf, unsigned short v) const {
        if (on) {
            *reinterpret_cast<unsigned short *>(pbuf + (p * 16 + k) * 2) = v;
            *reinterpret_cast<unsigned short *>(tbuf + (k * F1_PP + p) * 2) = v;
        }
    }
};
__device__ __forceinline__ void fe1_lds_init(char *lds, int Fo) {      // zeros everywhere, 1.0 in slot 9 of the real positions (both buffers)
    for (int i = threadIdx.x; i < 2 * (F1_PBYTES + F1_TBYTES) / 4; i += FB_THREADS) reinterpret_cast<unsigned *>(lds)[i] = 0u;
    __syncthreads();
    if (threadIdx.x < 2 * F1_PP) {
        const int buf = threadIdx.x / F1_PP, p = threadIdx.x - buf * F1_PP;
        if (p < Fo) {
            *reinterpret_cast<unsigned short *>(lds + buf * F1_PBYTES + (p * 16 + 9) * 2) = 0x3F80;
            *reinterpret_cast<unsigned short *>(lds + 2 * F1_PBYTES + buf * F1_TBYTES + (9 * F1_PP + p) * 2) = 0x3F80;
        }
    }
}

template <int NPB>
__global__ __launch_bounds__(FB_THREADS, 4) void fe_block1_fwd_mfma_kernel(const FeBlockArgs A) {
    extern __shared__ __attribute__((aligned(16))) char f1_lds[];
    char *Pb = f1_lds, *Tb = f1_lds + 2 * F1_PBYTES;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    const int Fo = A.Fo, N = Fo * FB_C;
    const float inv_n = 1.f / (float)N;
    bf16x8 wy, wr;
    fe1_filters(A, wave, lane, A.tmode == 1 ? 5 : 4, wy, wr);
    int off[NPB];
    bool ok[NPB];
    f32x4 g1[NPB], be1[NPB], g2[NPB], be2[NPB];
    const unsigned cidx0 = (unsigned)((pl * FB_C + 16 * wave + 4 * g) >> 1) * 0x9E3779B9u;      // pair index of (position pl, first channel) times the stream's stride
#pragma unroll
    for (int t = 0; t < NPB; ++t) {
        const int p = 16 * t + pl;
        ok[t] = p < Fo;
        off[t] = min(p, Fo - 1) * FB_C + 16 * wave + 4 * g;
        g1[t] = *reinterpret_cast<const f32x4 *>(A.g1 + off[t]);
        be1[t] = *reinterpret_cast<const f32x4 *>(A.be1 + off[t]);
        g2[t] = *reinterpret_cast<const f32x4 *>(A.g2 + off[t]);
        be2[t] = *reinterpret_cast<const f32x4 *>(A.be2 + off[t]);
    }
    const unsigned long long sd = A.seed_dev ? *A.seed_dev : 0ull;
    const unsigned thr1 = drop_thr16(A.p1), thr2 = drop_thr16(A.p2), thr1s = thr1 << 16, thr2s = thr2 << 16;
    const DropKey k1 = drop_key(A.seed1 + sd), k2 = drop_key(A.seed2 + sd);
    const float ks1 = drop_scale16(thr1), ks2 = drop_scale16(thr2);
    // LayerNorm statistics WITHOUT touching the 2 x 12 accumulators: y[p][c] = sum_k taps[p][k] W[k][c] (slot 9 = 1: the bias), so over a row
    //   sum y   = sum_k (sum_c W[k][c]) G[k][9] ,   sum y^2 = sum_{k,k'} (sum_c W[k][c] W[k'][c]) G[k][k'] ,   G = taps^T taps (16 x 16 per row),
    // and G is two MFMAs of the T tile with itself (positions on K). Every wave forms G and the four sums (y and r) for itself - a lane holds
    // G[4g + j][pl] and the matching weight sums, 16 multiply-adds and one wave_sum4 - so the statistics need no pass over the accumulators,
    // no LDS round and no other wave (24 adds + 24 multiply-adds + two-level reduction before). Filter sums in fp32 from the fp32 filters.
    float gy[4], gr[4], ay[4], ar[4];
    {
        // (the filters as [branch][slot][channel] rows in LDS - the tile area is not in use yet)
        float *wf = reinterpret_cast<float *>(f1_lds);      // [2][16][FB_C + 4] floats
        constexpr int WS = FB_C + 4;
        static_assert(2 * 16 * WS * 4 <= F1_LDS, "filter staging fits the kernel's LDS");
        const int ctr = A.tmode == 1 ? 5 : 4;
        for (int i = threadIdx.x; i < 16 * FB_C; i += FB_THREADS) {
            const int k = i / FB_C, c = i - k * FB_C;
            wf[k * WS + c] = k < 9 ? A.w1[c * 9 + k] : (k == 9 ? A.b1[c] : 0.f);
            wf[(16 + k) * WS + c] = k == ctr ? A.w2[c] : (k == 9 ? A.b2[c] : 0.f);
        }
        __syncthreads();
        // one (slot, slot') pair and half of the channels per thread; the partial sums meet in LDS behind the filters
        float *gw = wf + 2 * 16 * WS;       // [half][branch][16][16] dots, then [half][branch][16] row sums
        static_assert((2 * 16 * WS + 2 * 2 * 256 + 2 * 2 * 16) * 4 <= F1_LDS, "filter staging fits the kernel's LDS");
        {
            const int k = (threadIdx.x >> 4) & 15, k2 = threadIdx.x & 15, half = threadIdx.x >> 8;
            const float *ra = wf + k * WS + 64 * half, *rb = wf + k2 * WS + 64 * half;
            float dy = 0.f, sy = 0.f, dr = 0.f, sr = 0.f;
#pragma unroll 4
            for (int c = 0; c < 64; c += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(ra + c), z = *reinterpret_cast<const float4 *>(rb + c);
                const float4 xr = *reinterpret_cast<const float4 *>(ra + 16 * WS + c), zr = *reinterpret_cast<const float4 *>(rb + 16 * WS + c);
                dy += x.x * z.x + x.y * z.y + x.z * z.z + x.w * z.w;
                sy += (x.x + x.y) + (x.z + x.w);
                dr += xr.x * zr.x + xr.y * zr.y + xr.z * zr.z + xr.w * zr.w;
                sr += (xr.x + xr.y) + (xr.z + xr.w);
            }
            gw[(half * 2 + 0) * 256 + k * 16 + k2] = dy;
            gw[(half * 2 + 1) * 256 + k * 16 + k2] = dr;
            if (k2 == 0) { gw[1024 + (half * 2 + 0) * 16 + k] = sy; gw[1024 + (half * 2 + 1) * 16 + k] = sr; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * g + j;
            gy[j] = gw[k * 16 + pl] + gw[512 + k * 16 + pl];
            gr[j] = gw[256 + k * 16 + pl] + gw[768 + k * 16 + pl];
            ay[j] = pl == 9 ? gw[1024 + k] + gw[1024 + 32 + k] : 0.f;      // column 9 of G = the tap sums
            ar[j] = pl == 9 ? gw[1024 + 16 + k] + gw[1024 + 48 + k] : 0.f;
        }
        __syncthreads();
    }
    Fe1Stage stg;
    stg.init(A);
    fe1_lds_init(f1_lds, Fo);
    int b = (int)(blockIdx.x / (unsigned)A.To), to = (int)(blockIdx.x % (unsigned)A.To);
    const int stride_b = (int)(gridDim.x / (unsigned)A.To), stride_t = (int)(gridDim.x % (unsigned)A.To);
    __syncthreads();
    if ((long long)blockIdx.x < A.R) stg.store(Pb, Tb, stg.load(A, b, to));
    __syncthreads();
    int par = 0;
    // fragment addresses inside a T tile: positions 8g .. 8g+7 of slot pl, and positions 32 + 8g .. (g < 2; the other lanes read slot 15's zeros)
    const int tf0 = (pl * F1_PP + 8 * g) * 2, tf1 = g < 2 ? (pl * F1_PP + 32 + 8 * g) * 2 : (15 * F1_PP) * 2;
    for (long long row = blockIdx.x; row < A.R; row += gridDim.x, par ^= 1) {
        to += stride_t; b += stride_b;
        if (to >= A.To) { to -= A.To; ++b; }
        const bool more = row + gridDim.x < A.R;
        unsigned short nxt = 0;
        if (more) nxt = stg.load(A, b, to);                  // the next row's tap of this thread: in flight under the MFMAs and the statistics
        const char *P = Pb + par * F1_PBYTES, *Tt = Tb + par * F1_TBYTES;
        f32x4 y[NPB], r[NPB];
#pragma unroll
        for (int t = 0; t < NPB; ++t) {
            const bf16x8 taps = *reinterpret_cast<const bf16x8 *>(P + (16 * t + pl) * 32 + (g & 1) * 16);
            y[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wy, taps, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            r[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr, taps, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        f32x4 G;
        {
            const bf16x8 t0 = *reinterpret_cast<const bf16x8 *>(Tt + tf0);
            G = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, t0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (NPB > 2) {
                const bf16x8 t1 = *reinterpret_cast<const bf16x8 *>(Tt + tf1);
                G = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, t1, G, 0, 0, 0);
            }
        }
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0] = __builtin_fmaf(ay[j], G[j], s[0]); s[1] = __builtin_fmaf(ar[j], G[j], s[1]);
            s[2] = __builtin_fmaf(gy[j], G[j], s[2]); s[3] = __builtin_fmaf(gr[j], G[j], s[3]);
        }
        wave_sum4(s);
        if (more) stg.store(Pb + (par ^ 1) * F1_PBYTES, Tb + (par ^ 1) * F1_TBYTES, nxt);      // read last in the previous row, before its barrier
        __syncthreads();        // publishes the next row's taps (the statistics no longer need it)
        const float mu_y = s[0] * inv_n, mu_r = s[1] * inv_n;
        const float rs_y = rsqrtf(fmaxf(s[2] * inv_n - mu_y * mu_y, 0.f) + A.eps), rs_r = rsqrtf(fmaxf(s[3] * inv_n - mu_r * mu_r, 0.f) + A.eps);
        if (threadIdx.x == 0) *reinterpret_cast<float4 *>(A.stats + row * 4) = make_float4(mu_y, rs_y, mu_r, rs_r);
        const float cy = -mu_y * rs_y, cr = -mu_r * rs_r;
        const unsigned rk1 = drop_hash((unsigned long long)row, k1) + cidx0, rk2 = drop_hash((unsigned long long)row, k2) + cidx0;
        bf16_t *outr = (bf16_t *)A.out + row * N;
#pragma unroll
        for (int t = 0; t < NPB; ++t) {
            float o[4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                // (p = 0: thr = 0, every comparison holds and ks = 1 - no branch on the rate)
                const unsigned cth = (unsigned)(16 * t * (FB_C / 2) + h2) * 0x9E3779B9u;
                const unsigned wa = fe1_word(rk1, cth), wb = fe1_word(rk2, cth);
                const float m1[2] = {(wa << 16) >= thr1s ? ks1 : 0.f, wa >= thr1s ? ks1 : 0.f};
                const float m2[2] = {(wb << 16) >= thr2s ? ks2 : 0.f, wb >= thr2s ? ks2 : 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * h2 + e;
                    const float hy = __builtin_fmaf(y[t][j], rs_y, cy);
                    const float z = __builtin_fmaf(hy, g1[t][j], be1[t][j]);
                    const float act = fmaxf(z, z * A.slope);
                    const float hr = __builtin_fmaf(r[t][j], rs_r, cr);
                    const float rn = __builtin_fmaf(hr, g2[t][j], be2[t][j]);
                    o[j] = __builtin_fmaf(act, m1[e], rn) * m2[e];
                }
            }
            if (ok[t]) {
                bf16x4 ob;
                ob[0] = (bf16_t)o[0]; ob[1] = (bf16_t)o[1]; ob[2] = (bf16_t)o[2]; ob[3] = (bf16_t)o[3];
                *reinterpret_cast<bf16x4 *>(outr + off[t]) = ob;
            }
        }
    }
}

template <int NPB>
__global__ __launch_bounds__(FB_THREADS) void fe_block1_bwd_mfma_kernel(const FeBlockArgs A) {
    extern __shared__ __attribute__((aligned(16))) char f1_lds[];
    char *Pb = f1_lds, *Tb = f1_lds + 2 * F1_PBYTES;
    float *red = reinterpret_cast<float *>(f1_lds + 2 * (F1_PBYTES + F1_TBYTES));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    const int Fo = A.Fo, N = Fo * FB_C, c = 16 * wave + pl, centre = A.tmode == 1 ? 5 : 4;
    const float inv_n = 1.f / (float)N;
    bf16x8 wy, wr;
    fe1_filters(A, wave, lane, centre, wy, wr);
    // this lane: channel c at positions 16t + 4g + j
    int off[NPB];                        // element offset of (position 16t + 4g, c), clamped to the last real position
    float okf[4];                        // 1 / 0: position 16(NPB-1) + 4g + j is real
    float g1[NPB][4], be1[NPB][4], g2[NPB][4];
    float ag1[NPB][4], abe1[NPB][4], ag2[NPB][4], abe2[NPB][4];
    const unsigned cidx0 = (unsigned)((4 * g * FB_C + c) >> 1) * 0x9E3779B9u;      // pair index of (position 4g, c) times the stream's stride
#pragma unroll
    for (int t = 0; t < NPB; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = min(16 * t + 4 * g + j, Fo - 1), e = p * FB_C + c;
            if (j == 0) off[t] = e;
            g1[t][j] = A.g1[e]; be1[t][j] = A.be1[e]; g2[t][j] = A.g2[e];
            ag1[t][j] = abe1[t][j] = ag2[t][j] = abe2[t][j] = 0.f;
            if (t == NPB - 1) okf[j] = (16 * t + 4 * g + j) < Fo ? 1.f : 0.f;
        }
    }
    const int sh = (lane & 1) ? 0 : 16;          // even channel: low half of the pair's word
    f32x4 dwy = {0.f, 0.f, 0.f, 0.f}, dwr = {0.f, 0.f, 0.f, 0.f};      // D[slot 4g + j][c]
    const unsigned long long sd = A.seed_dev ? *A.seed_dev : 0ull;
    const unsigned thr1 = drop_thr16(A.p1), thr2 = drop_thr16(A.p2), thr1s = thr1 << 16, thr2s = thr2 << 16;
    const DropKey k1 = drop_key(A.seed1 + sd), k2 = drop_key(A.seed2 + sd);
    const float ks1 = drop_scale16(thr1), ks2 = drop_scale16(thr2);
    Fe1Stage stg;
    stg.init(A);
    fe1_lds_init(f1_lds, Fo);
    int b = (int)(blockIdx.x / (unsigned)A.To), to = (int)(blockIdx.x % (unsigned)A.To);
    const int stride_b = (int)(gridDim.x / (unsigned)A.To), stride_t = (int)(gridDim.x % (unsigned)A.To);
    __syncthreads();
    if ((long long)blockIdx.x < A.R) stg.store(Pb, Tb, stg.load(A, b, to));
    __syncthreads();
    int par = 0;
    for (long long row = blockIdx.x; row < A.R; row += gridDim.x, par ^= 1) {
        to += stride_t; b += stride_b;
        if (to >= A.To) { to -= A.To; ++b; }
        const bool more = row + gridDim.x < A.R;
        unsigned short nxt = 0;
        if (more) nxt = stg.load(A, b, to);
        const unsigned short *dor = reinterpret_cast<const unsigned short *>(A.dout) + row * N;
        const float4 st = *reinterpret_cast<const float4 *>(A.stats + row * 4);
        const float rs_y = st.y, rs_r = st.w, cy = -st.x * st.y, cr = -st.z * st.w;
        float d[NPB][4];
#pragma unroll
        for (int t = 0; t < NPB; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = min(16 * t + 4 * g + j, Fo - 1);
                d[t][j] = __uint_as_float((unsigned)dor[p * FB_C + c] << 16);
            }
        const char *P = Pb + par * F1_PBYTES, *T = Tb + par * F1_TBYTES;
        f32x4 y[NPB], r[NPB];
#pragma unroll
        for (int t = 0; t < NPB; ++t) {
            const bf16x8 taps = *reinterpret_cast<const bf16x8 *>(P + (16 * t + pl) * 32 + (g & 1) * 16);
            y[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(taps, wy, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            r[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(taps, wr, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // taps^T as the filter-gradient A operand: lane (slot pl, group g), k = 8g + e <-> position 4g + e (e < 4) of block 0 / 16 + 4g + e - 4 of block 1
        bf16x8 ta, tb;
        {
            const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(T + (pl * F1_PP + 4 * g) * 2);
            bf16x4 hi = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (NPB > 1) hi = *reinterpret_cast<const bf16x4 *>(T + (pl * F1_PP + 16 + 4 * g) * 2);
            ta = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf16x4 l2 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (NPB > 2) l2 = *reinterpret_cast<const bf16x4 *>(T + (pl * F1_PP + 32 + 4 * g) * 2);
            tb = (bf16x8){l2[0], l2[1], l2[2], l2[3], (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
        const unsigned rb1 = drop_hash((unsigned long long)row, k1) + cidx0, rb2 = drop_hash((unsigned long long)row, k2) + cidx0;
        float hy[NPB][4], hr[NPB][4], gy[NPB][4], gr[NPB][4];
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NPB; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr unsigned step = 64u * 0x9E3779B9u;      // the pair index grows by FB_C / 2 per position
                const unsigned cij = (unsigned)(16 * t + j) * step;
                const float m1 = (fe1_word(rb1, cij) << sh) >= thr1s ? ks1 : 0.f;
                const float m2 = (fe1_word(rb2, cij) << sh) >= thr2s ? ks2 : 0.f;
                float ds = d[t][j] * m2;
                if (t == NPB - 1) ds *= okf[j];
                const float h = __builtin_fmaf(y[t][j], rs_y, cy);
                const float z = __builtin_fmaf(h, g1[t][j], be1[t][j]);
                const float da = ds * m1;
                const float dz = z > 0.f ? da : da * A.slope;
                hy[t][j] = h;
                ag1[t][j] = __builtin_fmaf(dz, h, ag1[t][j]);
                abe1[t][j] += dz;
                const float gyv = dz * g1[t][j];
                gy[t][j] = gyv;
                s[0] += gyv; s[1] = __builtin_fmaf(gyv, h, s[1]);
                const float hrv = __builtin_fmaf(r[t][j], rs_r, cr);
                hr[t][j] = hrv;
                ag2[t][j] = __builtin_fmaf(ds, hrv, ag2[t][j]);
                abe2[t][j] += ds;
                const float grv = ds * g2[t][j];
                gr[t][j] = grv;
                s[2] += grv; s[3] = __builtin_fmaf(grv, hrv, s[3]);
            }
        if (more) stg.store(Pb + (par ^ 1) * F1_PBYTES, Tb + (par ^ 1) * F1_TBYTES, nxt);
        fb_wg_sum<4>(s, red + par * FB_WAVES * 4, wave, lane);
        const float m1y = s[0] * inv_n, m2y = s[1] * inv_n, m1r = s[2] * inv_n, m2r = s[3] * inv_n;
        float dyv[NPB][4], drv[NPB][4];
#pragma unroll
        for (int t = 0; t < NPB; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dyv[t][j] = rs_y * (gy[t][j] - m1y - hy[t][j] * m2y);
                drv[t][j] = rs_r * (gr[t][j] - m1r - hr[t][j] * m2r);
                if (t == NPB - 1) { dyv[t][j] *= okf[j]; drv[t][j] *= okf[j]; }      // (a padded position: gy = 0 but -m1y is not)
            }
        const bf16_t z0 = (bf16_t)0.f;
        bf16x8 by, br;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            by[j] = (bf16_t)dyv[0][j]; br[j] = (bf16_t)drv[0][j];
            by[4 + j] = NPB > 1 ? (bf16_t)dyv[NPB > 1 ? 1 : 0][j] : z0;
            br[4 + j] = NPB > 1 ? (bf16_t)drv[NPB > 1 ? 1 : 0][j] : z0;
        }
        dwy = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, by, dwy, 0, 0, 0);
        dwr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, br, dwr, 0, 0, 0);
        if (NPB > 2) {
            bf16x8 by2, br2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                by2[j] = (bf16_t)dyv[NPB > 2 ? 2 : 0][j]; br2[j] = (bf16_t)drv[NPB > 2 ? 2 : 0][j];
                by2[4 + j] = z0; br2[4 + j] = z0;
            }
            dwy = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tb, by2, dwy, 0, 0, 0);
            dwr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tb, br2, dwr, 0, 0, 0);
        }
    }
    // partial parameter gradients of this workgroup: [dw1 C*9 | db1 C | dw2 C | db2 C] then [dg1 | dbe1 | dg2 | dbe2]
    const int conv_w = FB_C * 12;
    float *mine = A.slab + (size_t)blockIdx.x * (conv_w + 4 * N);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = 4 * g + j;
        if (slot < 9) mine[c * 9 + slot] = dwy[j];
        if (slot == 9) { mine[FB_C * 9 + c] = dwy[j]; mine[FB_C * 11 + c] = dwr[j]; }
        if (slot == centre) mine[FB_C * 10 + c] = dwr[j];
    }
#pragma unroll
    for (int t = 0; t < NPB; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * t + 4 * g + j >= Fo) continue;
            const int e = off[t] + j * FB_C;
            mine[conv_w + e] = ag1[t][j];
            mine[conv_w + N + e] = abe1[t][j];
            mine[conv_w + 2 * N + e] = ag2[t][j];
            mine[conv_w + 3 * N + e] = abe2[t][j];
        }
}

int fb_out_len(int n) { return (n - 1) / 2 + 1; }
int fb_np(int Fo) { return (Fo + FB_WAVES - 1) / FB_WAVES; }
unsigned fb_grid(long long R, int per_cu) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const long long g = (long long)cus * per_cu;
    return (unsigned)(R < g ? R : g);
}
constexpr int FB_BWD_WGS_PER_CU = 1, FB_BWD_MAX_WGS = 512;

// persistent grid: as many workgroups as are resident at once (1 or 2 per CU, by the instantiation's register count)
template <typename K>
unsigned fb_resident_grid(K kern, long long R, size_t lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, FB_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    return fb_grid(R, per_cu > 2 ? 2 : per_cu);
}

// block 1 in bf16 on the matrix-core kernels (TSASR_FE_MFMA=0: the vector kernels above, for A/B runs)
bool fe1_enabled() {
    static const bool on = [] { const char *e = getenv("TSASR_FE_MFMA"); return !e || e[0] != '0'; }();
    return on;
}
bool fe1_takes(bool conv, int io_dtype, int Fo) { return conv && io_dtype == TSASR_BF16 && Fo <= F1_PP && fe1_enabled(); }
void fe1_launch_fwd(int Fo, hipStream_t st, const FeBlockArgs &a) {
#define F1_FWD(NPBV)                                                                   \
    {                                                                                  \
        auto kern = fe_block1_fwd_mfma_kernel<NPBV>;                                   \
        static const unsigned grid_full = fb_resident_grid(kern, 1ll << 40, F1_LDS);   \
        kern<<<(unsigned)(a.R < grid_full ? a.R : grid_full), FB_THREADS, F1_LDS, st>>>(a); \
    }
    if (Fo <= 16) F1_FWD(1) else if (Fo <= 32) F1_FWD(2) else F1_FWD(3)
#undef F1_FWD
}
void fe1_launch_bwd(int Fo, unsigned grid, hipStream_t st, const FeBlockArgs &a) {
    if (Fo <= 16) fe_block1_bwd_mfma_kernel<1><<<grid, FB_THREADS, F1_LDS, st>>>(a);
    else if (Fo <= 32) fe_block1_bwd_mfma_kernel<2><<<grid, FB_THREADS, F1_LDS, st>>>(a);
    else fe_block1_bwd_mfma_kernel<3><<<grid, FB_THREADS, F1_LDS, st>>>(a);
}

template <typename T, bool CONV>
void fb_launch_fwd(int np, bool full, hipStream_t st, const FeBlockArgs &a) {
#define FB_FWD(NPV, FULLV)                                                            \
    {                                                                                 \
        auto kern = fe_block_fwd_kernel<T, CONV, NPV, FULLV>;                         \
        static const unsigned grid_full = fb_resident_grid(kern, 1ll << 40, 0);       \
        kern<<<(unsigned)(a.R < grid_full ? a.R : grid_full), FB_THREADS, 0, st>>>(a); \
    }
    switch (np) {
    case 1: if (full) FB_FWD(1, true) else FB_FWD(1, false) break;
    case 2: if (full) FB_FWD(2, true) else FB_FWD(2, false) break;
    case 3: if (full) FB_FWD(3, true) else FB_FWD(3, false) break;
    case 4: if (full) FB_FWD(4, true) else FB_FWD(4, false) break;
    default: if (full) FB_FWD(5, true) else FB_FWD(5, false) break;
    }
#undef FB_FWD
}
template <typename T, bool CONV>
void fb_launch_bwd(int np, bool full, unsigned grid, hipStream_t st, const FeBlockArgs &a) {
    const size_t lds = (size_t)(2 * FB_WAVES * 4 + (CONV ? FB_WAVES * FB_C * 12 : 0)) * sizeof(float);
#define FB_BWD(NPV, FULLV)                                                                                                       \
    {                                                                                                                            \
        auto kern = fe_block_bwd_kernel<T, CONV, NPV, FULLV>;                                                                    \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        kern<<<grid, FB_THREADS, lds, st>>>(a);                                                                                  \
    }
    switch (np) {
    case 1: if (full) FB_BWD(1, true) else FB_BWD(1, false) break;
    case 2: if (full) FB_BWD(2, true) else FB_BWD(2, false) break;
    case 3: if (full) FB_BWD(3, true) else FB_BWD(3, false) break;
    case 4: if (full) FB_BWD(4, true) else FB_BWD(4, false) break;
    default: if (full) FB_BWD(5, true) else FB_BWD(5, false) break;
    }
#undef FB_BWD
}

}  // namespace

extern "C" {

/* 1 when the fused ConvBlock kernels cover a block with C output channels over F' output frequency positions. */
int tsasr_frontend_block_supported(int Fo, int C) { return C == FB_C && Fo >= 1 && Fo <= FB_WAVES * FB_MAXNP; }

/* Fused ConvBlock forward.
 *   x != NULL (block 1, C_in = 1): x [B,T,F] features, filters w1 [C,1,3,3] (kernel axes (F,T)), b1, w2 [C], b2; y1 = y2 = NULL.
 *   x == NULL (wider blocks):      y1, y2 [B*T', F', C] = the two convolution outputs (bias included); T, F are still the INPUT sizes.
 * g1/be1, g2/be2 fp32 [F'*C]: LayerNorm affine of the 3x3 branch / the 1x1 branch. out [B,T',F',C]; stats fp32 [B*T'][4].
 * Dropout: p_inner on the activated 3x3 branch, p_outer on the sum; mask = f(seed + *seed_dev, element index). */
int tsasr_frontend_block_fwd(const void *x, const void *y1, const void *y2, const float *w1, const float *b1, const float *w2,
                             const float *b2, const float *g1, const float *be1, const float *g2, const float *be2, void *out,
                             float *stats, int B, int T, int F, int C, int causal, float slope, float eps, float p_inner,
                             unsigned long long seed_inner, float p_outer, unsigned long long seed_outer,
                             const unsigned long long *seed_dev, int io_dtype, void *stream) {
    const bool conv = x != nullptr;
    TSASR_CHECK_ARG(g1 && be1 && g2 && be2 && out && stats, "tsasr_frontend_block_fwd: null pointer");
    TSASR_CHECK_ARG(conv ? (w1 && b1 && w2 && b2 && !y1 && !y2) : (y1 && y2), "tsasr_frontend_block_fwd: pass either x and the filters, or y1 and y2");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2, "tsasr_frontend_block_fwd: bad shape (B=%d T=%d F=%d)", B, T, F);
    const int To = fb_out_len(T), Fo = fb_out_len(F);
    TSASR_CHECK_ARG(tsasr_frontend_block_supported(Fo, C), "tsasr_frontend_block_fwd: unsupported block (F'=%d C=%d; need C=%d, F'<=%d)", Fo, C, FB_C, FB_WAVES * FB_MAXNP);
    TSASR_CHECK_ARG(p_inner >= 0.f && p_inner < 1.f && p_outer >= 0.f && p_outer < 1.f, "tsasr_frontend_block_fwd: bad dropout");
    TSASR_CHECK_ARG(slope >= 0.f && slope <= 1.f, "tsasr_frontend_block_fwd: LeakyReLU slope %f outside [0, 1]", slope);
    TSASR_CHECK_ARG((long long)B * To < (1ll << 31), "tsasr_frontend_block_fwd: too many rows");
    FeBlockArgs a{};
    a.x = x; a.y1 = y1; a.y2 = y2; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.g1 = g1; a.be1 = be1; a.g2 = g2; a.be2 = be2;
    a.out = out; a.stats = stats; a.R = (long long)B * To; a.Tn = T; a.F = F; a.To = To; a.Fo = Fo;
    a.tmode = causal ? 1 : 0; a.fmode = causal ? 2 : 0;
    a.slope = slope; a.eps = eps; a.p1 = p_inner; a.p2 = p_outer; a.seed1 = seed_inner; a.seed2 = seed_outer; a.seed_dev = seed_dev;
    hipStream_t st = (hipStream_t)stream;
    const int np = fb_np(Fo);
    const bool full = Fo == np * FB_WAVES;
    if (fe1_takes(conv, io_dtype, Fo)) fe1_launch_fwd(Fo, st, a);
    else if (io_dtype == TSASR_BF16) { if (conv) fb_launch_fwd<bf16_t, true>(np, full, st, a); else fb_launch_fwd<bf16_t, false>(np, full, st, a); }
    else if (io_dtype == TSASR_F32) { if (conv) fb_launch_fwd<float, true>(np, full, st, a); else fb_launch_fwd<float, false>(np, full, st, a); }
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_frontend_block_fwd");
    return 0;
}

/* floats in dparams: with_conv ? C*12 + 4*F'*C : 4*F'*C */
size_t tsasr_frontend_block_dparams(int Fo, int C, int with_conv) { return (size_t)(with_conv ? C * 12 : 0) + (size_t)4 * Fo * C; }
size_t tsasr_frontend_block_bwd_workspace_bytes(int Fo, int C, int with_conv) {
    return align_up((size_t)FB_BWD_MAX_WGS * tsasr_frontend_block_dparams(Fo, C, with_conv) * sizeof(float), 256);
}

/* Fused ConvBlock backward (same arguments and seeds as the forward call; stats as written by it).
 * dparams fp32, overwritten: [dw1 C*9 | db1 C | dw2 C | db2 C] (only when x != NULL) then [dg1 | dbe1 | dg2 | dbe2] (F'*C each).
 * x == NULL: dy1, dy2 [B*T', F', C] receive the gradients of the two convolution outputs. */
int tsasr_frontend_block_bwd(const void *x, const void *y1, const void *y2, const void *dout, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *g1, const float *be1, const float *g2,
                             const float *stats, void *dy1, void *dy2, float *dparams, int B, int T, int F, int C, int causal,
                             float slope, float p_inner, unsigned long long seed_inner, float p_outer,
                             unsigned long long seed_outer, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                             size_t workspace_bytes, void *stream) {
    const bool conv = x != nullptr;
    TSASR_CHECK_ARG(dout && g1 && be1 && g2 && stats && dparams && workspace, "tsasr_frontend_block_bwd: null pointer");
    TSASR_CHECK_ARG(conv ? (w1 && b1 && w2 && b2) : (y1 && y2 && dy1 && dy2), "tsasr_frontend_block_bwd: pass either x and the filters, or y1, y2, dy1, dy2");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2, "tsasr_frontend_block_bwd: bad shape (B=%d T=%d F=%d)", B, T, F);
    const int To = fb_out_len(T), Fo = fb_out_len(F);
    TSASR_CHECK_ARG(tsasr_frontend_block_supported(Fo, C), "tsasr_frontend_block_bwd: unsupported block (F'=%d C=%d)", Fo, C);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_frontend_block_bwd_workspace_bytes(Fo, C, conv), "tsasr_frontend_block_bwd: workspace too small");
    FeBlockArgs a{};
    a.x = x; a.y1 = y1; a.y2 = y2; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.g1 = g1; a.be1 = be1; a.g2 = g2;
    a.stats = const_cast<float *>(stats); a.dout = dout; a.dy1 = dy1; a.dy2 = dy2; a.slab = (float *)workspace;
    a.R = (long long)B * To; a.Tn = T; a.F = F; a.To = To; a.Fo = Fo;
    a.tmode = causal ? 1 : 0; a.fmode = causal ? 2 : 0;
    a.slope = slope; a.p1 = p_inner; a.p2 = p_outer; a.seed1 = seed_inner; a.seed2 = seed_outer; a.seed_dev = seed_dev;
    unsigned grid = fb_grid(a.R, FB_BWD_WGS_PER_CU);
    if (grid > (unsigned)FB_BWD_MAX_WGS) grid = FB_BWD_MAX_WGS;
    hipStream_t st = (hipStream_t)stream;
    const int np = fb_np(Fo);
    const bool full = Fo == np * FB_WAVES;
    if (fe1_takes(conv, io_dtype, Fo)) fe1_launch_bwd(Fo, grid, st, a);
    else if (io_dtype == TSASR_BF16) { if (conv) fb_launch_bwd<bf16_t, true>(np, full, grid, st, a); else fb_launch_bwd<bf16_t, false>(np, full, grid, st, a); }
    else if (io_dtype == TSASR_F32) { if (conv) fb_launch_bwd<float, true>(np, full, grid, st, a); else fb_launch_bwd<float, false>(np, full, grid, st, a); }
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    const int width = (int)tsasr_frontend_block_dparams(Fo, C, conv);
    tsasr_reduce_submit((const float *)workspace, dparams, width, (int)grid, width, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_frontend_block_bwd");
    return 0;
}

}  // extern "C"
