// HBM-bound row kernels of the Conformer block for gfx950: LayerNorm (+fused LeakyReLU) fwd/bwd,
// bias+LeakyReLU+dropout fwd/bwd, dropout+scale+residual(+time mask) fwd/bwd, column-sum reduction of partials.
//
// Replaces the ATen chains the reference issues for (SB = vendor/speechbrain/speechbrain):
//   nn.LayerNorm / SB LayerNorm                      SB/nnet/normalization.py:172-223, Conformer.py:73,93,194-214,216-217
//   Linear bias + activation() + Dropout             SB/nnet/attention.py:820-836 (PositionalwiseFeedForward)
//   Dropout + 0.5*x + residual, masked_fill_ of pads Conformer.py:113-114,239-259
// Design: every kernel is one pass, 16-byte (8 x bf16 / 4 x fp32) accesses per lane, fp32 math, one wave per row for
// model-width rows (D <= 2048) and one workgroup per row for the front-end's [F*C] rows. Column reductions
// (dgamma/dbeta/dbias) are two-stage and deterministic: per-workgroup partial rows, then a small column-sum kernel -
// no float atomics. Dropout is counter-based (hash of element index and a per-call seed): the mask is never stored,
// backward regenerates it.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

template <typename T> struct Vec;  // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };

template <typename T, int N> __device__ __forceinline__ void ldv(const T *p, float (&o)[N]);
template <> __device__ __forceinline__ void ldv<float, 4>(const float *p, float (&o)[4]) {
    const float4 a = *reinterpret_cast<const float4 *>(p);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
template <> __device__ __forceinline__ void ldv<bf16_t, 8>(const bf16_t *p, float (&o)[8]) { ld8(p, o); }
template <> __device__ __forceinline__ void ldv<float, 8>(const float *p, float (&o)[8]) { ld8(p, o); }   // fp32 parameters beside bf16 rows
template <typename T, int N> __device__ __forceinline__ void stv(T *p, const float (&v)[N]);
template <> __device__ __forceinline__ void stv<float, 4>(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void stv<bf16_t, 8>(bf16_t *p, const float (&v)[8]) { st8(p, v); }

// block-wide sum for TPR (threads per row) = 64 (one wave) or 256 (one workgroup)
template <int TPR> __device__ __forceinline__ float row_sum(float v, float *red) {
    if (TPR == 32) return half_wave_sum(v);   // two rows per wave (256 bf16 columns = 32 lanes x 16 bytes: a whole wave per row left half of it idle)
    v = wave_sum(v);
    if (TPR == 64) return v;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm forward: y = act(gamma * (x - mean) * rstd + beta)
// ---------------------------------------------------------------------------------------------------
template <typename T, int TPR, int ITERS>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, T *__restrict__ y,
                                                            float *__restrict__ mean, float *__restrict__ rstd, long long M,
                                                            int D, float eps, float slope) {
    constexpr int N = Vec<T>::N;
    __shared__ float red[4];
    const int rows_per_blk = 256 / TPR;
    long long row = (long long)blockIdx.x * rows_per_blk + threadIdx.x / TPR;
    const int l = threadIdx.x % TPR;
    if (TPR == 64 && row >= M) return;
    const bool row_valid = row < M;               // TPR == 32: the other half of the wave may still own a row - no early exit
    if (TPR == 32 && !row_valid) row = M - 1;
    const T *xr = x + row * D;
    float v[ITERS][N];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * TPR + l) * N;
        if (c < D) {
            ldv<T, N>(xr + c, v[it]);
#pragma unroll
            for (int j = 0; j < N; ++j) s += v[it][j];
        }
    }
    const float mu = row_sum<TPR>(s, red) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * TPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const float d = v[it][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(row_sum<TPR>(q, red) / D + eps);
    if (TPR == 32 && !row_valid) return;
    if (l == 0) { mean[row] = mu; rstd[row] = rs; }
    T *yr = y + row * D;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * TPR + l) * N;
        if (c < D) {
            float g[N], b[N], o[N];
            ldv<float, 4>(gamma + c, *reinterpret_cast<float(*)[4]>(&g[0]));
            ldv<float, 4>(beta + c, *reinterpret_cast<float(*)[4]>(&b[0]));
            if (N == 8) {
                ldv<float, 4>(gamma + c + 4, *reinterpret_cast<float(*)[4]>(&g[N == 8 ? 4 : 0]));
                ldv<float, 4>(beta + c + 4, *reinterpret_cast<float(*)[4]>(&b[N == 8 ? 4 : 0]));
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = (v[it][j] - mu) * rs * g[j] + b[j];
                if (slope >= 0.f) t = lrelu(t, slope);
                o[j] = t;
            }
            stv<T, N>(yr + c, o);
        }
    }
}

// LayerNorm backward. Each workgroup walks `rows_per_wg` consecutive rows, keeps column partials of dgamma/dbeta in
// registers, and writes them to part[blockIdx][2][D]; colsum_kernel finishes.
template <typename T, int TPR, int ITERS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            T *__restrict__ dx, float *__restrict__ part, long long M, int D,
                                                            float slope, int rows_per_wg, const T *__restrict__ dadd) {
    constexpr int N = Vec<T>::N;
    __shared__ float red[4];
    __shared__ float colred[(TPR == 64) ? 1 : 1];
    (void)colred;
    const int rpb = 256 / TPR;
    const int sub = threadIdx.x / TPR, l = threadIdx.x % TPR;
    float ag[ITERS][N], abt[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int j = 0; j < N; ++j) ag[it][j] = abt[it][j] = 0.f;
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    // TPR == 32: the two halves of a wave share the loop bound (row_sum is made of wave-wide instructions); the odd half may idle
    for (int rb = (TPR == 32 ? (sub & ~1) : sub); rb < rows_per_wg; rb += rpb) {
        const int rr = TPR == 32 ? rb + (sub & 1) : rb;
        const long long row = r0 + rr;
        const bool live = row < M && rr < rows_per_wg;
        if (TPR == 64 && !live) break;
        const T *xr = x + row * D, *dyr = dy + row * D;
        const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
        float xh[ITERS][N], gdy[ITERS][N];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * TPR + l) * N;
            if (c < D && live) {
                float xv[N], dv[N], g[N], b[N];
                ldv<T, N>(xr + c, xv);
                ldv<T, N>(dyr + c, dv);
                ldv<float, 4>(gamma + c, *reinterpret_cast<float(*)[4]>(&g[0]));
                if (N == 8) ldv<float, 4>(gamma + c + 4, *reinterpret_cast<float(*)[4]>(&g[N == 8 ? 4 : 0]));
                if (slope >= 0.f) {
                    ldv<float, 4>(beta + c, *reinterpret_cast<float(*)[4]>(&b[0]));
                    if (N == 8) ldv<float, 4>(beta + c + 4, *reinterpret_cast<float(*)[4]>(&b[N == 8 ? 4 : 0]));
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float h = (xv[j] - mu) * rs;
                    float d = dv[j];
                    if (slope >= 0.f && (h * g[j] + b[j]) <= 0.f) d *= slope;  // LeakyReLU' on the pre-activation
                    xh[it][j] = h;
                    ag[it][j] += d * h;
                    abt[it][j] += d;
                    const float gd = d * g[j];
                    gdy[it][j] = gd;
                    s1 += gd;
                    s2 += gd * h;
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) xh[it][j] = gdy[it][j] = 0.f;
            }
        }
        const float m1 = row_sum<TPR>(s1, red) / D;
        const float m2 = row_sum<TPR>(s2, red) / D;
        if (live) {
            T *dxr = dx + row * D;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (it * TPR + l) * N;
                if (c < D) {
                    float o[N], ad[N];
                    if (dadd) ldv<T, N>(dadd + row * D + c, ad);     // gradient that reached x along another path (residual): summed here
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = rs * (gdy[it][j] - m1 - xh[it][j] * m2) + (dadd ? ad[j] : 0.f);
                    stv<T, N>(dxr + c, o);
                }
            }
        }
    }
    // column partials: combine the `rpb` row-slots of this workgroup through LDS, then one row per workgroup
    extern __shared__ __attribute__((aligned(16))) float colbuf[];  // [rpb][2][D] when rpb > 1
    float *pw = part + (size_t)blockIdx.x * 2 * D;
    if (rpb == 1) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * TPR + l) * N;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < N; ++j) { pw[c + j] = ag[it][j]; pw[D + c + j] = abt[it][j]; }
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * TPR + l) * N;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    colbuf[(sub * 2 + 0) * D + c + j] = ag[it][j];
                    colbuf[(sub * 2 + 1) * D + c + j] = abt[it][j];
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * D; i += 256) {
            float s = 0.f;
            for (int q = 0; q < rpb; ++q) s += colbuf[q * 2 * D + i];
            pw[i] = s;
        }
    }
}

// LayerNorm backward for WIDE rows (one workgroup per row at a time: the [F, C] = 5120 / 2560-element rows of the convolutional
// front-end, 164 MB per tensor). The narrow-row kernel above, used here, ran at 1.8 TB/s: per row it re-read gamma and beta (40 KB
// of fp32 parameters from L2 beside 20 KB of data), issued every load under a column guard (serialized, csrc/attention.hip note) and
// exposed a full memory round trip plus four barriers per row. Here the parameters live in registers for the whole row loop, loads
// are unconditional (clamped column), the next row (and its mean / rstd) is requested before this row's reductions, and the two row
// sums share one exchange whose buffer alternates between rows (one barrier per row).
template <typename T, int N> __device__ __forceinline__ void unpack16(const uint4 &w, float (&o)[N]);
template <> __device__ __forceinline__ void unpack16<bf16_t, 8>(const uint4 &w, float (&o)[8]) {
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { o[2 * q] = __uint_as_float(u[q] << 16); o[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack16<float, 4>(const uint4 &w, float (&o)[4]) {
    o[0] = __uint_as_float(w.x); o[1] = __uint_as_float(w.y); o[2] = __uint_as_float(w.z); o[3] = __uint_as_float(w.w);
}

template <typename T, int ITERS>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                 T *__restrict__ dx, float *__restrict__ part, long long M, int D,
                                                                 float slope, int rows_per_wg) {
    constexpr int N = Vec<T>::N;
    __shared__ float red[2][2][4];
    const int l = threadIdx.x, w = l >> 6;
    int cc[ITERS];
    bool ok[ITERS];
    float g[ITERS][N], b[ITERS][N], ag[ITERS][N], abt[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 256 + l) * N;
        ok[it] = c < D;
        cc[it] = ok[it] ? c : 0;
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            ldv<float, 4>(gamma + cc[it] + j, *reinterpret_cast<float(*)[4]>(&g[it][j]));
            ldv<float, 4>(beta + cc[it] + j, *reinterpret_cast<float(*)[4]>(&b[it][j]));
        }
#pragma unroll
        for (int j = 0; j < N; ++j) ag[it][j] = abt[it][j] = 0.f;
    }
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
    uint4 xn[ITERS], dn[ITERS];
    float mu_n = 0.f, rs_n = 0.f;
    auto request = [&](long long row) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            xn[it] = *reinterpret_cast<const uint4 *>(x + row * D + cc[it]);
            dn[it] = *reinterpret_cast<const uint4 *>(dy + row * D + cc[it]);
        }
        mu_n = mean[row];
        rs_n = rstd[row];
    };
    if (r0 < r1) request(r0);
    for (long long row = r0; row < r1; ++row) {
        float xh[ITERS][N], gdy[ITERS][N];
        const float mu = mu_n, rs = rs_n;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            float xv[N], dv[N];
            unpack16<T, N>(xn[it], xv);
            unpack16<T, N>(dn[it], dv);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float h = (xv[j] - mu) * rs;
                float d = ok[it] ? dv[j] : 0.f;
                if (slope >= 0.f && (h * g[it][j] + b[it][j]) <= 0.f) d *= slope;  // LeakyReLU' on the pre-activation
                xh[it][j] = h;
                ag[it][j] += d * h;
                abt[it][j] += d;
                const float gd = d * g[it][j];
                gdy[it][j] = gd;
                s1 += gd;
                s2 += gd * h;
            }
        }
        if (row + 1 < r1) request(row + 1);       // in flight during the reductions and the stores of this row
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const int par = (int)(row & 1);
        if ((l & 63) == 0) { red[par][0][w] = s1; red[par][1][w] = s2; }
        __syncthreads();
        const float m1 = ((red[par][0][0] + red[par][0][1]) + (red[par][0][2] + red[par][0][3])) / D;
        const float m2 = ((red[par][1][0] + red[par][1][1]) + (red[par][1][2] + red[par][1][3])) / D;
        T *dxr = dx + row * D;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (ok[it]) {
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = rs * (gdy[it][j] - m1 - xh[it][j] * m2);
                stv<T, N>(dxr + cc[it], o);
            }
        }
    }
    float *pw = part + (size_t)blockIdx.x * 2 * D;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if (ok[it]) {
#pragma unroll
            for (int j = 0; j < N; j += 4) {
                *reinterpret_cast<float4 *>(pw + cc[it] + j) = make_float4(ag[it][j], ag[it][j + 1], ag[it][j + 2], ag[it][j + 3]);
                *reinterpret_cast<float4 *>(pw + D + cc[it] + j) = make_float4(abt[it][j], abt[it][j + 1], abt[it][j + 2], abt[it][j + 3]);
            }
        }
    }
}

// LayerNorm forward for wide rows: the same treatment (parameters in registers across a row loop, unconditional loads, next row in
// flight during this row's two reductions; mean, then centred sum of squares - two passes over registers, not over memory).
template <typename T, int ITERS>
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, T *__restrict__ y,
                                                                 float *__restrict__ mean, float *__restrict__ rstd, long long M, int D,
                                                                 float eps, float slope, int rows_per_wg) {
    constexpr int N = Vec<T>::N;
    __shared__ float red[2][2][4];
    const int l = threadIdx.x, w = l >> 6;
    int cc[ITERS];
    bool ok[ITERS];
    float g[ITERS][N], b[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 256 + l) * N;
        ok[it] = c < D;
        cc[it] = ok[it] ? c : 0;
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            ldv<float, 4>(gamma + cc[it] + j, *reinterpret_cast<float(*)[4]>(&g[it][j]));
            ldv<float, 4>(beta + cc[it] + j, *reinterpret_cast<float(*)[4]>(&b[it][j]));
        }
    }
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
    uint4 xn[ITERS];
    auto request = [&](long long row) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) xn[it] = *reinterpret_cast<const uint4 *>(x + row * D + cc[it]);
    };
    if (r0 < r1) request(r0);
    for (long long row = r0; row < r1; ++row) {
        float v[ITERS][N];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            unpack16<T, N>(xn[it], v[it]);
#pragma unroll
            for (int j = 0; j < N; ++j) s += ok[it] ? v[it][j] : 0.f;
        }
        if (row + 1 < r1) request(row + 1);
        const int par = (int)(row & 1);
        s = wave_sum(s);
        if ((l & 63) == 0) red[par][0][w] = s;
        __syncthreads();
        const float mu = ((red[par][0][0] + red[par][0][1]) + (red[par][0][2] + red[par][0][3])) / D;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
#pragma unroll
            for (int j = 0; j < N; ++j) { const float d = ok[it] ? v[it][j] - mu : 0.f; q += d * d; }
        q = wave_sum(q);
        if ((l & 63) == 0) red[par][1][w] = q;
        __syncthreads();
        const float rs = rsqrtf(((red[par][1][0] + red[par][1][1]) + (red[par][1][2] + red[par][1][3])) / D + eps);
        if (l == 0) { mean[row] = mu; rstd[row] = rs; }
        T *yr = y + row * D;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (ok[it]) {
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float t = (v[it][j] - mu) * rs * g[it][j] + b[it][j];
                    if (slope >= 0.f) t = lrelu(t, slope);
                    o[j] = t;
                }
                stv<T, N>(yr + cc[it], o);
            }
        }
    }
}

// Column sums of an [M, N] io-dtype matrix (bias gradient of a GEMM-shaped layer whose output gradient is already in HBM):
// per-workgroup partial rows part[wg][N]; the caller finishes with tsasr_reduce_submit. 16-byte loads, 256/(N/8) rows per pass.
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T *__restrict__ x, float *__restrict__ part, long long M, int Ncols,
                                                          int rows_per_wg) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float cred[];   // [slots][Ncols]
    const int chunks = Ncols / N, tpr = chunks < 256 ? chunks : 256, slots = 256 / tpr;
    const int slot = threadIdx.x / tpr, lane = threadIdx.x % tpr;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
    for (int c = lane * N; c < Ncols; c += tpr * N) {
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        if (slot < slots) {
            long long row = r0 + slot;
            for (; row + 3 * slots < r1; row += 4 * slots) {       // four independent row loads in flight
                float v[4][N];
#pragma unroll
                for (int q = 0; q < 4; ++q) ldv<T, N>(x + (row + (long long)q * slots) * Ncols + c, v[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < N; ++j) acc[j] += v[q][j];
            }
            for (; row < r1; row += slots) {
                float v[N];
                ldv<T, N>(x + row * Ncols + c, v);
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] += v[j];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) cred[slot * Ncols + c + j] = acc[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ncols; i += 256) {
        float s = 0.f;
        for (int q = 0; q < slots; ++q) s += cred[q * Ncols + i];
        part[(size_t)blockIdx.x * Ncols + i] = s;
    }
}

// out[c] = sum_n part[n][c]: 16 columns x 16 row-slices per workgroup, slices combined through LDS (fixed order ->
// deterministic). part rows are `stride` wide; columns [0,D) go to out_a, [D,stride) to out_b.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ part, float *__restrict__ out_a,
                                                     float *__restrict__ out_b, int nparts, int D, int stride) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < stride) {
        int n = slice;
        for (; n + 48 < nparts; n += 64) {  // 4 independent loads in flight per lane
            s0 += part[(size_t)n * stride + col];
            s1 += part[(size_t)(n + 16) * stride + col];
            s2 += part[(size_t)(n + 32) * stride + col];
            s3 += part[(size_t)(n + 48) * stride + col];
        }
        for (; n < nparts; n += 16) s0 += part[(size_t)n * stride + col];
    }
    red[slice][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && col < stride) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        if (col < D) { if (out_a) out_a[col] = s; }
        else if (out_b) out_b[col - D] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// y = dropout(act(x + bias))           (act = LeakyReLU(slope) when slope >= 0)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bias_act_dropout_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias,
                                                                   T *__restrict__ y, long long total, int Ncols, float slope,
                                                                   float p, unsigned long long seed, const unsigned long long *__restrict__ seed_dev) {
    constexpr int N = Vec<T>::N;
    if (seed_dev) seed += *seed_dev;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float keep_scale = drop_scale16(thr);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * N; i < total; i += (long long)gridDim.x * 256 * N) {
        float v[N];
        ldv<T, N>(x + i, v);
        const int c = (int)(i % Ncols);
        const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)i, dk, thr) : ~0u;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float t = v[j] + (bias ? bias[c + j] : 0.f);
            if (slope >= 0.f) t = lrelu(t, slope);
            if (p > 0.f) t = ((km >> j) & 1u) ? t * keep_scale : 0.f;
            v[j] = t;
        }
        stv<T, N>(y + i, v);
    }
}

// dx = dy * keep/(1-p) * act'(y) ; dbias partial rows part[blockIdx][Ncols]. A workgroup owns `rows_per_wg` consecutive
// rows; its 256 threads are laid out as (256 / tpr) row slots x tpr column chunks so that narrow rows still use every lane;
// the row slots are combined through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_dropout_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ y,
                                                                   T *__restrict__ dx, float *__restrict__ part, long long M,
                                                                   int Ncols, float slope, float p, unsigned long long seed,
                                                                   const unsigned long long *__restrict__ seed_dev, int rows_per_wg) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float cred[];  // [slots][Ncols] when slots > 1 and part != NULL
    if (seed_dev) seed += *seed_dev;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float keep_scale = drop_scale16(thr);
    const int chunks = Ncols / N, tpr = chunks < 256 ? chunks : 256, slots = 256 / tpr;
    const int slot = threadIdx.x / tpr, lane = threadIdx.x % tpr;
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    for (int c = lane * N; c < Ncols; c += tpr * N) {
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        if (slot < slots) {
#pragma unroll 4
            for (int rr = slot; rr < rows_per_wg; rr += slots) {
                const long long row = r0 + rr;
                if (row >= M) break;
                const long long i = row * Ncols + c;
                float d[N], yv[N];
                ldv<T, N>(dy + i, d);
                if (slope >= 0.f) ldv<T, N>(y + i, yv);
                const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)i, dk, thr) : ~0u;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float g = d[j];
                    if (p > 0.f) g = ((km >> j) & 1u) ? g * keep_scale : 0.f;
                    if (slope >= 0.f && yv[j] < 0.f) g *= slope;  // sign(y) = sign(pre-activation) for kept elements
                    d[j] = g;
                    acc[j] += g;
                }
                stv<T, N>(dx + i, d);
            }
        }
        if (part) {
            if (slots == 1) {
                if (slot == 0)
#pragma unroll
                for (int j = 0; j < N; ++j) part[(size_t)blockIdx.x * Ncols + c + j] = acc[j];
            } else if (slot < slots) {
#pragma unroll
                for (int j = 0; j < N; ++j) cred[slot * Ncols + c + j] = acc[j];
            }
        }
    }
    if (part && slots > 1) {
        __syncthreads();
        for (int i = threadIdx.x; i < Ncols; i += 256) {
            float s = 0.f;
            for (int q = 0; q < slots; ++q) s += cred[q * Ncols + i];
            part[(size_t)blockIdx.x * Ncols + i] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// out = res + alpha * mask_t( dropout(x + bias) )      rows = [B, T] flattened; mask_t zeroes rows t >= valid_lens[b]
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias,
                                                              const T *__restrict__ res, T *__restrict__ out, long long total,
                                                              int Ncols, float alpha, float p, unsigned long long seed,
                                                              const unsigned long long *__restrict__ seed_dev,
                                                              const int32_t *__restrict__ valid_lens, int Trows, float p2,
                                                              unsigned long long seed2) {
    constexpr int N = Vec<T>::N;
    if (seed_dev) { seed += *seed_dev; seed2 += *seed_dev; }
    const unsigned thr = drop_thr16(p), thr2 = drop_thr16(p2);
    const DropKey dk = drop_key(seed), dk2 = drop_key(seed2);
    const float keep_scale = drop_scale16(thr), keep_scale2 = drop_scale16(thr2);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * N; i < total; i += (long long)gridDim.x * 256 * N) {
        float v[N], r[N];
        ldv<T, N>(x + i, v);
        if (res) ldv<T, N>(res + i, r);
        const long long row = i / Ncols;
        const int c = (int)(i - row * Ncols);
        const bool live = valid_lens ? ((int)(row % Trows) < valid_lens[row / Trows]) : true;
        const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)i, dk, thr) : ~0u;
        const unsigned km2 = p2 > 0.f ? drop_keep_mask<N>((unsigned long long)i, dk2, thr2) : ~0u;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float t = v[j] + (bias ? bias[c + j] : 0.f);
            if (p > 0.f) t = ((km >> j) & 1u) ? t * keep_scale : 0.f;
            t = live ? t * alpha : 0.f;
            t += res ? r[j] : 0.f;
            if (p2 > 0.f) {                                  // outer dropout of the sum; the intermediate is rounded to the io type
                if (sizeof(T) == 2) t = (float)(bf16_t)t;     // first, exactly as a separate second pass would have read it
                t = ((km2 >> j) & 1u) ? t * keep_scale2 : 0.f;
            }
            v[j] = t;
        }
        stv<T, N>(out + i, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_add_bwd_kernel(const T *__restrict__ dout, T *__restrict__ dx,
                                                              float *__restrict__ part, long long M, int Ncols, float alpha,
                                                              float p, unsigned long long seed,
                                                              const unsigned long long *__restrict__ seed_dev,
                                                              const int32_t *__restrict__ valid_lens, int Trows, int rows_per_wg,
                                                              float p2, unsigned long long seed2, T *__restrict__ dres) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float cred[];
    if (seed_dev) { seed += *seed_dev; seed2 += *seed_dev; }
    const unsigned thr = drop_thr16(p), thr2 = drop_thr16(p2);
    const DropKey dk = drop_key(seed), dk2 = drop_key(seed2);
    const float keep_scale = drop_scale16(thr), keep_scale2 = drop_scale16(thr2);
    const int chunks = Ncols / N, tpr = chunks < 256 ? chunks : 256, slots = 256 / tpr;
    const int slot = threadIdx.x / tpr, lane = threadIdx.x % tpr;
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    for (int c = lane * N; c < Ncols; c += tpr * N) {
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        if (slot < slots) {
#pragma unroll 4
            for (int rr = slot; rr < rows_per_wg; rr += slots) {
                const long long row = r0 + rr;
                if (row >= M) break;
                const long long i = row * Ncols + c;
                const bool live = valid_lens ? ((int)(row % Trows) < valid_lens[row / Trows]) : true;
                float d[N];
                ldv<T, N>(dout + i, d);
                const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)i, dk, thr) : ~0u;
                if (p2 > 0.f) {                               // gradient through the outer dropout = gradient of the residual input
                    const unsigned km2 = drop_keep_mask<N>((unsigned long long)i, dk2, thr2);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float g2 = ((km2 >> j) & 1u) ? d[j] * keep_scale2 : 0.f;
                        if (sizeof(T) == 2) g2 = (float)(bf16_t)g2;
                        d[j] = g2;
                    }
                    if (dres) stv<T, N>(dres + i, d);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float g = live ? d[j] * alpha : 0.f;
                    if (p > 0.f) g = ((km >> j) & 1u) ? g * keep_scale : 0.f;
                    d[j] = g;
                    acc[j] += g;
                }
                stv<T, N>(dx + i, d);
            }
        }
        if (part) {
            if (slots == 1) {
                if (slot == 0)
#pragma unroll
                for (int j = 0; j < N; ++j) part[(size_t)blockIdx.x * Ncols + c + j] = acc[j];
            } else if (slot < slots) {
#pragma unroll
                for (int j = 0; j < N; ++j) cred[slot * Ncols + c + j] = acc[j];
            }
        }
    }
    if (part && slots > 1) {
        __syncthreads();
        for (int i = threadIdx.x; i < Ncols; i += 256) {
            float s = 0.f;
            for (int q = 0; q < slots; ++q) s += cred[q * Ncols + i];
            part[(size_t)blockIdx.x * Ncols + i] = s;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Fused residual tail + LayerNorm (the seam between two Conformer sub-blocks):
//   s = res + alpha * timemask(dropout_p(x + bias)) ;  y = LayerNorm(s) * gamma + beta        (one wave per row)
// replaces Dropout -> (0.5*)x + residual -> [masked_fill_] -> nn.LayerNorm, Conformer.py:113-114,239-259 + :73,194-217.
// Backward: d_s = LayerNorm_bwd(dy) + dout (gradient arriving through the residual path);  dres = d_s ;
//           dx = alpha * timemask * dropmask/(1-p) * d_s ;  column partials for dgamma, dbeta, dbias.
// ---------------------------------------------------------------------------------------------------
template <typename T, int ITERS, bool HW = false>   // HW: two rows per wave (D <= 32 lanes x 16 bytes)
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias,
                                                                const T *__restrict__ res, T *__restrict__ s_out, T *__restrict__ y,
                                                                float *__restrict__ mean, float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                long long M, int D, float alpha, float p, unsigned long long seed,
                                                                const unsigned long long *__restrict__ seed_dev,
                                                                const int32_t *__restrict__ valid_lens, int Trows, float eps) {
    constexpr int N = Vec<T>::N, LPR = HW ? 32 : 64;   // lanes per row
    long long row = HW ? ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x >> 5) & 1) : (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & (LPR - 1);
    if (!HW && row >= M) return;
    const bool row_valid = row < M;                    // HW: no early exit, the other half-wave may own a row (loads clamped, stores guarded)
    if (HW && !row_valid) row = M - 1;
    // ONE round trip: every operand - the device seed, the utterance length, the parameters, the row - is requested up front,
    // unconditionally (optional ones from a stand-in address, columns clamped) and masked afterwards. `if (seed_dev) seed += *seed_dev`,
    // `valid_lens ? valid_lens[..] : ..` and `if (bias) load` are guarded loads: each was waited for where it stood, three dependent
    // round trips in front of the row's own.
    const bool has_bias = bias != nullptr, has_vl = valid_lens != nullptr;
    const unsigned long long *seed_p = seed_dev ? seed_dev : reinterpret_cast<const unsigned long long *>(gamma);
    const int32_t *vl_p = has_vl ? valid_lens : reinterpret_cast<const int32_t *>(gamma);
    const float *bias_p = has_bias ? bias : gamma;
    const int trows = has_vl ? max(Trows, 1) : 1;
    const unsigned long long seed_add = *seed_p;
    const int vl = vl_p[has_vl ? row / trows : 0];
    float v[ITERS][N];
    float gv[ITERS][N], bt[ITERS][N];      // requested with the row, used after the two reductions
    float xv[ITERS][N], rv[ITERS][N], bv[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = min((it * LPR + l) * N, D - N);
        ldv<float, N>(gamma + c, gv[it]);
        ldv<float, N>(beta + c, bt[it]);
        ldv<T, N>(x + row * D + c, xv[it]);
        ldv<T, N>(res + row * D + c, rv[it]);
        ldv<float, N>(bias_p + c, bv[it]);
    }
    if (seed_dev) seed += seed_add;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float ks = drop_scale16(thr);
    const bool live = !has_vl || ((int)(row % trows) < vl);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
            const unsigned long long idx = (unsigned long long)row * D + c;
            const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)idx, dk, thr) : ~0u;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = xv[it][j] + (has_bias ? bv[it][j] : 0.f);
                if (p > 0.f) t = ((km >> j) & 1u) ? t * ks : 0.f;
                t = live ? t * alpha : 0.f;
                t += rv[it][j];
                if (sizeof(T) == 2) t = (float)(bf16_t)t;   // statistics of the STORED (rounded) row, as a separate LN would see
                v[it][j] = t;
                sum += t;
            }
            if (row_valid) stv<T, N>(s_out + row * D + c, v[it]);
        }
    }
    const float mu = (HW ? half_wave_sum(sum) : wave_sum(sum)) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const float d = v[it][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf((HW ? half_wave_sum(q) : wave_sum(q)) / D + eps);
    if (!row_valid) return;
    if (l == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = (v[it][j] - mu) * rs * gv[it][j] + bt[it][j];
            stv<T, N>(y + row * D + c, o);
        }
    }
}

// part rows per workgroup: [dgamma D | dbeta D | dbias D]
template <typename T, int ITERS, bool HW = false>   // HW: two rows per wave, one per half (D <= 32 lanes x 16 bytes)
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ dout,
                                                                const T *__restrict__ s_in, const float *__restrict__ gamma,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                T *__restrict__ dres, T *__restrict__ dx, float *__restrict__ part,
                                                                long long M, int D, float alpha, float p, unsigned long long seed,
                                                                const unsigned long long *__restrict__ seed_dev,
                                                                const int32_t *__restrict__ valid_lens, int Trows, int rows_per_wg) {
    constexpr int N = Vec<T>::N, LPR = HW ? 32 : 64, RPP = HW ? 8 : 4;   // lanes per row; rows per workgroup pass
    extern __shared__ __attribute__((aligned(16))) float colbuf[];  // [RPP row slots][3][D]
    if (seed_dev) seed += *seed_dev;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float ks = drop_scale16(thr);
    const int slot = threadIdx.x / LPR, l = threadIdx.x & (LPR - 1);
    float ag[ITERS][N], abt[ITERS][N], abx[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int j = 0; j < N; ++j) ag[it][j] = abt[it][j] = abx[it][j] = 0.f;
    // this lane's columns (clamped: every load below is issued unconditionally and masked afterwards), gamma in registers
    int cc[ITERS];
    bool ok[ITERS];
    float gm[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        ok[it] = c < D;
        cc[it] = ok[it] ? c : 0;
        ldv<float, N>(gamma + cc[it], gm[it]);
    }
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
    // a row slot (wave, or half-wave) walks rows r0 + slot, +RPP, ...: the next row (s, dy, dout, mean, rstd) is requested before this row's reductions
    float sn[ITERS][N], dn[ITERS][N], on[ITERS][N], mu_n = 0.f, rs_n = 0.f;
    // optional operands are ALWAYS requested, from a stand-in address when absent (and ignored): `if (ptr) load` is a guarded load - it was
    // waited for on the spot, and drained every other request of the row with it; the same for the utterance length read per row
    const bool has_dout = dout != nullptr, has_vl = valid_lens != nullptr;
    const T *dout_p = has_dout ? dout : dy;
    const int32_t *vl_p = has_vl ? valid_lens : reinterpret_cast<const int32_t *>(mean);
    const int trows = has_vl ? max(Trows, 1) : 1;
    int vl_n = 0;
    auto request = [&](long long row) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            ldv<T, N>(s_in + row * D + cc[it], sn[it]);
            ldv<T, N>(dy + row * D + cc[it], dn[it]);
            ldv<T, N>(dout_p + row * D + cc[it], on[it]);
        }
        mu_n = mean[row];
        rs_n = rstd[row];
        vl_n = vl_p[row / trows];
    };
    // HW: both halves of a wave stay in the loop together (the reductions are wave-wide instructions); a half without a row
    // re-reads the last row and contributes nothing
    if (HW ? r0 + (slot & ~1) < r1 : r0 + slot < r1) request(min(r0 + slot, r1 - 1));
    for (long long row_w = r0 + (HW ? (slot & ~1) : slot); row_w < r1; row_w += RPP) {
        const long long row_u = HW ? row_w + (slot & 1) : row_w;
        const bool row_valid = row_u < r1;
        const long long row = row_valid ? row_u : r1 - 1;
        const float mu = mu_n, rs = rs_n;
        const bool live = !has_vl || ((int)(row % trows) < vl_n);
        float xh[ITERS][N], gd[ITERS][N], dov[ITERS][N];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool use = HW ? (ok[it] && row_valid) : ok[it];
                const float h = use ? (sn[it][j] - mu) * rs : 0.f, dv = use ? dn[it][j] : 0.f;
                xh[it][j] = h;
                ag[it][j] += dv * h;
                abt[it][j] += dv;
                const float g = dv * gm[it][j];
                gd[it][j] = g;
                s1 += g;
                s2 += g * h;
                dov[it][j] = has_dout ? on[it][j] : 0.f;
            }
        }
        if (row_w + RPP < r1) request(min(row_u + RPP, r1 - 1));
        const float m1 = (HW ? half_wave_sum(s1) : wave_sum(s1)) / D, m2 = (HW ? half_wave_sum(s2) : wave_sum(s2)) / D;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (HW ? (ok[it] && row_valid) : ok[it]) {
                const int c = cc[it];
                float ds[N], dxv[N];
                const unsigned long long idx = (unsigned long long)row * D + c;
                const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)idx, dk, thr) : ~0u;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float d = rs * (gd[it][j] - m1 - xh[it][j] * m2) + dov[it][j];
                    ds[j] = d;
                    float g = live ? d * alpha : 0.f;
                    if (p > 0.f) g = ((km >> j) & 1u) ? g * ks : 0.f;
                    dxv[j] = g;
                    abx[it][j] += g;
                }
                stv<T, N>(dres + row * D + c, ds);
                stv<T, N>(dx + row * D + c, dxv);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                colbuf[(slot * 3 + 0) * D + c + j] = ag[it][j];
                colbuf[(slot * 3 + 1) * D + c + j] = abt[it][j];
                colbuf[(slot * 3 + 2) * D + c + j] = abx[it][j];
            }
        }
    }
    __syncthreads();
    float *pw = part + (size_t)blockIdx.x * 3 * D;
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        float a = colbuf[i] + colbuf[3 * D + i] + colbuf[6 * D + i] + colbuf[9 * D + i];
        if (HW) a += colbuf[12 * D + i] + colbuf[15 * D + i] + colbuf[18 * D + i] + colbuf[21 * D + i];
        pw[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------------
// The seam between two Conformer LAYERS is two LayerNorms in a row: norm2 of layer i (Conformer.py:259) and the first macaron FFN's
// LayerNorm of layer i+1 (Conformer.py:194-217; after the last layer: the encoder's final norm, models/conformer.py:233). One pass:
//   s = res + alpha * timemask(dropout_p(x + bias)) ;  y = LN(s) * gamma + beta ;  z = LN(y) * gamma2 + beta2
// with the statistics of z taken from the STORED (rounded) y, so (y, z) are bit-identical to add_layernorm + layernorm.
// Backward:  dy_total = LN_bwd2(dz) + dy (gradient that reaches y along the residual path; may be NULL), rounded to the io dtype as
// the tensor it replaces was; then exactly add_layernorm_bwd. y is recomputed from s (beta needed), never read.
// part rows per workgroup: [dgamma D | dbeta D | dbias D | dgamma2 D | dbeta2 D]
// ---------------------------------------------------------------------------------------------------
template <typename T, int ITERS, bool HW>
__global__ __launch_bounds__(256) void add_layernorm2_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias,
                                                                 const T *__restrict__ res, T *__restrict__ s_out, T *__restrict__ y,
                                                                 T *__restrict__ z, float *__restrict__ mean, float *__restrict__ rstd,
                                                                 float *__restrict__ mean2, float *__restrict__ rstd2,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ gamma2, const float *__restrict__ beta2,
                                                                 long long M, int D, float alpha, float p, unsigned long long seed,
                                                                 const unsigned long long *__restrict__ seed_dev,
                                                                 const int32_t *__restrict__ valid_lens, int Trows, float eps, float eps2) {
    constexpr int N = Vec<T>::N, LPR = HW ? 32 : 64;
    long long row = HW ? ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((threadIdx.x >> 5) & 1) : (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & (LPR - 1);
    if (!HW && row >= M) return;
    const bool row_valid = row < M;                    // HW: no early exit (the reductions are wave-wide instructions); stores guarded
    if (HW && !row_valid) row = M - 1;
    // one round trip for every operand (see add_layernorm_fwd_kernel): optional ones from a stand-in address, columns clamped
    const bool has_bias = bias != nullptr, has_vl = valid_lens != nullptr;
    const unsigned long long *seed_p = seed_dev ? seed_dev : reinterpret_cast<const unsigned long long *>(gamma);
    const int32_t *vl_p = has_vl ? valid_lens : reinterpret_cast<const int32_t *>(gamma);
    const float *bias_p = has_bias ? bias : gamma;
    const int trows = has_vl ? max(Trows, 1) : 1;
    const unsigned long long seed_add = *seed_p;
    const int vl = vl_p[has_vl ? row / trows : 0];
    float v[ITERS][N], gv[ITERS][N], bt[ITERS][N], gv2[ITERS][N], bt2[ITERS][N];
    float xv[ITERS][N], rv[ITERS][N], bv[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = min((it * LPR + l) * N, D - N);
        ldv<float, N>(gamma + c, gv[it]);
        ldv<float, N>(beta + c, bt[it]);
        ldv<float, N>(gamma2 + c, gv2[it]);
        ldv<float, N>(beta2 + c, bt2[it]);
        ldv<T, N>(x + row * D + c, xv[it]);
        ldv<T, N>(res + row * D + c, rv[it]);
        ldv<float, N>(bias_p + c, bv[it]);
    }
    if (seed_dev) seed += seed_add;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float ks = drop_scale16(thr);
    const bool live = !has_vl || ((int)(row % trows) < vl);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
            const unsigned long long idx = (unsigned long long)row * D + c;
            const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)idx, dk, thr) : ~0u;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = xv[it][j] + (has_bias ? bv[it][j] : 0.f);
                if (p > 0.f) t = ((km >> j) & 1u) ? t * ks : 0.f;
                t = live ? t * alpha : 0.f;
                t += rv[it][j];
                if (sizeof(T) == 2) t = (float)(bf16_t)t;
                v[it][j] = t;
                sum += t;
            }
            if (row_valid) stv<T, N>(s_out + row * D + c, v[it]);
        }
    }
    const float mu = (HW ? half_wave_sum(sum) : wave_sum(sum)) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const float d = v[it][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf((HW ? half_wave_sum(q) : wave_sum(q)) / D + eps);
    float sum2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = (v[it][j] - mu) * rs * gv[it][j] + bt[it][j];
                if (sizeof(T) == 2) t = (float)(bf16_t)t;       // the second LayerNorm sees the stored row
                v[it][j] = t;
                sum2 += t;
            }
            if (row_valid) stv<T, N>(y + row * D + c, v[it]);
        }
    }
    const float mu2 = (HW ? half_wave_sum(sum2) : wave_sum(sum2)) / D;
    float q2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const float d = v[it][j] - mu2; q2 += d * d; }
        }
    }
    const float rs2 = rsqrtf((HW ? half_wave_sum(q2) : wave_sum(q2)) / D + eps2);
    if (!row_valid) return;
    if (l == 0) { mean[row] = mu; rstd[row] = rs; mean2[row] = mu2; rstd2[row] = rs2; }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = (v[it][j] - mu2) * rs2 * gv2[it][j] + bt2[it][j];
            stv<T, N>(z + row * D + c, o);
        }
    }
}

template <typename T, int ITERS, bool HW>
__global__ __launch_bounds__(256) void add_layernorm2_bwd_kernel(const T *__restrict__ dz, const T *__restrict__ dy, const T *__restrict__ dout,
                                                                 const T *__restrict__ s_in, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, const float *__restrict__ gamma2,
                                                                 const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                 const float *__restrict__ mean2, const float *__restrict__ rstd2,
                                                                 T *__restrict__ dres, T *__restrict__ dx, float *__restrict__ part,
                                                                 long long M, int D, float alpha, float p, unsigned long long seed,
                                                                 const unsigned long long *__restrict__ seed_dev,
                                                                 const int32_t *__restrict__ valid_lens, int Trows, int rows_per_wg) {
    constexpr int N = Vec<T>::N, LPR = HW ? 32 : 64, RPP = HW ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) float colbuf[];  // [RPP row slots][5][D]
    if (seed_dev) seed += *seed_dev;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float ks = drop_scale16(thr);
    const int slot = threadIdx.x / LPR, l = threadIdx.x & (LPR - 1);
    float ag[ITERS][N], abt[ITERS][N], abx[ITERS][N], ag2[ITERS][N], abt2[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int j = 0; j < N; ++j) ag[it][j] = abt[it][j] = abx[it][j] = ag2[it][j] = abt2[it][j] = 0.f;
    int cc[ITERS];
    bool ok[ITERS];
    float gm[ITERS][N], bm[ITERS][N], gm2[ITERS][N];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        ok[it] = c < D;
        cc[it] = ok[it] ? c : 0;
        ldv<float, N>(gamma + cc[it], gm[it]);
        ldv<float, N>(beta + cc[it], bm[it]);
        ldv<float, N>(gamma2 + cc[it], gm2[it]);
    }
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, M);
    float sn[ITERS][N], zn[ITERS][N], dn[ITERS][N], on[ITERS][N], mu_n = 0.f, rs_n = 0.f, mu2_n = 0.f, rs2_n = 0.f;
    // optional operands always requested, from a stand-in address when absent (see add_layernorm_bwd_kernel)
    const bool has_dy = dy != nullptr, has_dout = dout != nullptr, has_vl = valid_lens != nullptr;
    const T *dy_p = has_dy ? dy : dz, *dout_p = has_dout ? dout : dz;
    const int32_t *vl_p = has_vl ? valid_lens : reinterpret_cast<const int32_t *>(mean);
    const int trows = has_vl ? max(Trows, 1) : 1;
    int vl_n = 0;
    auto request = [&](long long row) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            ldv<T, N>(s_in + row * D + cc[it], sn[it]);
            ldv<T, N>(dz + row * D + cc[it], zn[it]);
            ldv<T, N>(dy_p + row * D + cc[it], dn[it]);
            ldv<T, N>(dout_p + row * D + cc[it], on[it]);
        }
        mu_n = mean[row];
        rs_n = rstd[row];
        mu2_n = mean2[row];
        rs2_n = rstd2[row];
        vl_n = vl_p[row / trows];
    };
    if (HW ? r0 + (slot & ~1) < r1 : r0 + slot < r1) request(min(r0 + slot, r1 - 1));
    for (long long row_w = r0 + (HW ? (slot & ~1) : slot); row_w < r1; row_w += RPP) {
        const long long row_u = HW ? row_w + (slot & 1) : row_w;
        const bool row_valid = row_u < r1;
        const long long row = row_valid ? row_u : r1 - 1;
        const float mu = mu_n, rs = rs_n, mu2 = mu2_n, rs2 = rs2_n;
        const bool live = !has_vl || ((int)(row % trows) < vl_n);
        float xh[ITERS][N], xh2[ITERS][N], gd[ITERS][N], dov[ITERS][N], dyr[ITERS][N];
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool use = HW ? (ok[it] && row_valid) : ok[it];
                const float h = use ? (sn[it][j] - mu) * rs : 0.f;
                float yv = h * gm[it][j] + bm[it][j];
                if (sizeof(T) == 2) yv = (float)(bf16_t)yv;                    // the stored y of the forward
                const float h2 = use ? (yv - mu2) * rs2 : 0.f, dzv = use ? zn[it][j] : 0.f;
                xh[it][j] = h;
                xh2[it][j] = h2;
                ag2[it][j] += dzv * h2;
                abt2[it][j] += dzv;
                const float g2 = dzv * gm2[it][j];
                gd[it][j] = g2;
                t1 += g2;
                t2 += g2 * h2;
                dov[it][j] = has_dout ? on[it][j] : 0.f;
                dyr[it][j] = (has_dy && use) ? dn[it][j] : 0.f;
            }
        }
        if (row_w + RPP < r1) request(min(row_u + RPP, r1 - 1));
        const float n1 = (HW ? half_wave_sum(t1) : wave_sum(t1)) / D, n2 = (HW ? half_wave_sum(t2) : wave_sum(t2)) / D;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool use = HW ? (ok[it] && row_valid) : ok[it];
                float dv = rs2 * (gd[it][j] - n1 - xh2[it][j] * n2) + dyr[it][j];   // gradient of y: through the second LayerNorm + the residual path
                if (sizeof(T) == 2) dv = (float)(bf16_t)dv;
                dv = use ? dv : 0.f;
                const float h = xh[it][j];
                ag[it][j] += dv * h;
                abt[it][j] += dv;
                const float g = dv * gm[it][j];
                gd[it][j] = g;
                s1 += g;
                s2 += g * h;
            }
        }
        const float m1 = (HW ? half_wave_sum(s1) : wave_sum(s1)) / D, m2 = (HW ? half_wave_sum(s2) : wave_sum(s2)) / D;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (HW ? (ok[it] && row_valid) : ok[it]) {
                const int c = cc[it];
                float ds[N], dxv[N];
                const unsigned long long idx = (unsigned long long)row * D + c;
                const unsigned km = p > 0.f ? drop_keep_mask<N>((unsigned long long)idx, dk, thr) : ~0u;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float d = rs * (gd[it][j] - m1 - xh[it][j] * m2) + dov[it][j];
                    ds[j] = d;
                    float g = live ? d * alpha : 0.f;
                    if (p > 0.f) g = ((km >> j) & 1u) ? g * ks : 0.f;
                    dxv[j] = g;
                    abx[it][j] += g;
                }
                stv<T, N>(dres + row * D + c, ds);
                stv<T, N>(dx + row * D + c, dxv);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + l) * N;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                colbuf[(slot * 5 + 0) * D + c + j] = ag[it][j];
                colbuf[(slot * 5 + 1) * D + c + j] = abt[it][j];
                colbuf[(slot * 5 + 2) * D + c + j] = abx[it][j];
                colbuf[(slot * 5 + 3) * D + c + j] = ag2[it][j];
                colbuf[(slot * 5 + 4) * D + c + j] = abt2[it][j];
            }
        }
    }
    __syncthreads();
    float *pw = part + (size_t)blockIdx.x * 5 * D;
    for (int i = threadIdx.x; i < 5 * D; i += 256) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) a += colbuf[q * 5 * D + i];
        pw[i] = a;
    }
}

// out3[k][c] = sum_parts part[n][k*D + c], k = 0..2 (dgamma, dbeta, dbias) ; any out pointer may be NULL
__global__ __launch_bounds__(256) void colsum3_kernel(const float *__restrict__ part, float *__restrict__ o0, float *__restrict__ o1,
                                                      float *__restrict__ o2, int nparts, int D) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4, col = blockIdx.x * 16 + cl, W = 3 * D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < W) {
        int n = slice;
        for (; n + 48 < nparts; n += 64) {
            s0 += part[(size_t)n * W + col];
            s1 += part[(size_t)(n + 16) * W + col];
            s2 += part[(size_t)(n + 32) * W + col];
            s3 += part[(size_t)(n + 48) * W + col];
        }
        for (; n < nparts; n += 16) s0 += part[(size_t)n * W + col];
    }
    s0 += s1 + s2 + s3;
    red[slice][cl] = s0;
    __syncthreads();
    if (slice == 0 && col < W) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        float *o = col < D ? o0 : (col < 2 * D ? o1 : o2);
        if (o) o[col % D] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------
// (LayerNorm-family backward kernels: at least 16 rows per workgroup since round 2 - 8 gave twice the partial rows for the batched
//  reduction at the end of backward to read; measured 12.88 -> 12.76 ms per step, 32 rows: 12.95)
static int pick_rows_per_wg(long long M, int min_rows) {
    // ~1024 workgroups (4 per CU: a wave walks its rows one after the other, so other waves must cover its memory round
    // trips) unless rows are few; at least `min_rows` rows each so partial slabs stay small
    static const long long target = 1024;
    long long r = (M + target - 1) / target;
    static const int min_env = 0;
    if (min_env > 0) min_rows = min_env;
    if (r < min_rows) r = min_rows;
    return (int)r;
}

template <typename T>
static int launch_ln_fwd(const void *x, const float *g, const float *b, void *y, float *mean, float *rstd, long long M, int D,
                         float eps, float slope, hipStream_t st) {
    constexpr int N = Vec<T>::N;
    const int per_wave = 64 * N, per_wg = 256 * N;
#define LN_FWD(TPR, IT)                                                                                                  \
    layernorm_fwd_kernel<T, TPR, IT><<<(unsigned)((M + (256 / TPR) - 1) / (256 / TPR)), 256, 0, st>>>(                    \
        (const T *)x, g, b, (T *)y, mean, rstd, M, D, eps, slope)
    const int rpw = (int)std::max<long long>(4, (M + 1023) / 1024);      // wide rows: ~1024 workgroups walking rpw rows each
#define LN_FWD_WIDE(IT)                                                                                                   \
    layernorm_fwd_wide_kernel<T, IT><<<(unsigned)((M + rpw - 1) / rpw), 256, 0, st>>>((const T *)x, g, b, (T *)y, mean, rstd, M, D, eps, slope, rpw)
    static const int half_rows = 1;
    if (half_rows && D <= per_wave / 2) LN_FWD(32, 1);
    else if (D <= per_wave) LN_FWD(64, 1);
    else if (D <= 2 * per_wave) LN_FWD(64, 2);
    else if (D <= 4 * per_wave) LN_FWD(64, 4);
    else if (D <= 2 * per_wg) LN_FWD_WIDE(2);
    else if (D <= 3 * per_wg) LN_FWD_WIDE(3);
    else if (D <= 4 * per_wg) LN_FWD_WIDE(4);
    else if (D <= 6 * per_wg) LN_FWD_WIDE(6);
    else if (D <= 8 * per_wg) LN_FWD_WIDE(8);
    else return -1;
#undef LN_FWD_WIDE
#undef LN_FWD
    return 0;
}

template <typename T>
static int launch_ln_bwd(const void *dy, const void *x, const float *g, const float *b, const float *mean, const float *rstd,
                         void *dx, float *part, long long M, int D, float slope, int rpw, int nwg, hipStream_t st, const void *dadd = nullptr) {
    if (dadd && D > 4 * 64 * Vec<T>::N) return -3;      // the pass-through sum is implemented for the one-wave-per-row kernels
    constexpr int N = Vec<T>::N;
    const int per_wave = 64 * N, per_wg = 256 * N;
#define LN_BWD(TPR, IT)                                                                                                  \
    layernorm_bwd_kernel<T, TPR, IT><<<nwg, 256, (TPR <= 64 ? (size_t)(256 / TPR) * 2 * D * sizeof(float) : 0), st>>>(   \
        (const T *)dy, (const T *)x, g, b, mean, rstd, (T *)dx, part, M, D, slope, rpw, (const T *)dadd)
#define LN_BWD_WIDE(IT)                                                                                                   \
    layernorm_bwd_wide_kernel<T, IT><<<nwg, 256, 0, st>>>((const T *)dy, (const T *)x, g, b, mean, rstd, (T *)dx, part, M, D, slope, rpw)
    static const int half_rows = 1;
    if (half_rows && D <= per_wave / 2) LN_BWD(32, 1);
    else if (D <= per_wave) LN_BWD(64, 1);
    else if (D <= 2 * per_wave) LN_BWD(64, 2);
    else if (D <= 4 * per_wave) LN_BWD(64, 4);
    else if (D <= 2 * per_wg) LN_BWD_WIDE(2);
    else if (D <= 3 * per_wg) LN_BWD_WIDE(3);
    else if (D <= 4 * per_wg) LN_BWD_WIDE(4);
    else if (D <= 6 * per_wg) LN_BWD_WIDE(6);
    else if (D <= 8 * per_wg) LN_BWD_WIDE(8);
    else return -1;
#undef LN_BWD_WIDE
#undef LN_BWD
    return 0;
}

__global__ void seed_advance_kernel(unsigned long long *s, unsigned long long inc) { *s += inc; }

extern "C" {

/* *seed_dev += increment: one launch at the start of every training step (inside the captured graph), so that the
 * counter-based dropout masks change from step to step although every kernel argument is frozen by the capture. */
int tsasr_seed_advance(unsigned long long *seed_dev, unsigned long long increment, void *stream) {
    TSASR_CHECK_ARG(seed_dev, "tsasr_seed_advance: null pointer");
    seed_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(seed_dev, increment);
    TSASR_CHECK_LAUNCH("tsasr_seed_advance");
    return 0;
}

int tsasr_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                        long long M, int D, float eps, float act_slope, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && gamma && beta && y && mean && rstd, "tsasr_layernorm_fwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "tsasr_layernorm_fwd: bad shape M=%lld D=%d (D %% 8 == 0 required)", M, D);
    int rc = io_dtype == TSASR_F32 ? launch_ln_fwd<float>(x, gamma, beta, y, mean, rstd, M, D, eps, act_slope, (hipStream_t)stream)
           : io_dtype == TSASR_BF16 ? launch_ln_fwd<bf16_t>(x, gamma, beta, y, mean, rstd, M, D, eps, act_slope, (hipStream_t)stream)
                                    : -2;
    TSASR_CHECK_ARG(rc == 0, "tsasr_layernorm_fwd: D=%d too large or bad io_dtype %d", D, io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_layernorm_fwd");
    return 0;
}

size_t tsasr_layernorm_bwd_workspace_bytes(long long M, int D) {
    const int rpw = pick_rows_per_wg(M, 16);
    const long long nwg = (M + rpw - 1) / rpw;
    return align_up((size_t)nwg * 2 * D * sizeof(float), 256);
}

}  // extern "C"

static int layernorm_bwd_impl(const void *dy, const void *x, const float *gamma, const float *beta, const float *mean,
                              const float *rstd, void *dx, float *dgamma, float *dbeta, long long M, int D, float act_slope,
                              int io_dtype, void *workspace, size_t workspace_bytes, void *stream, const void *dadd) {
    TSASR_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && workspace, "tsasr_layernorm_bwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "tsasr_layernorm_bwd: bad shape");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_layernorm_bwd_workspace_bytes(M, D), "tsasr_layernorm_bwd: workspace too small");
    int rpw = pick_rows_per_wg(M, 16);
    const bool wide = D > 4 * 64 * (io_dtype == TSASR_BF16 ? 8 : 4);
    static const long long wide_wgs = 512;
    if (wide) rpw = (int)std::max<long long>(rpw, (M + wide_wgs - 1) / wide_wgs);   // wide-row kernel: two workgroups per CU, each prefetching its next row
    const int nwg = (int)((M + rpw - 1) / rpw);
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    int rc = io_dtype == TSASR_F32 ? launch_ln_bwd<float>(dy, x, gamma, beta, mean, rstd, dx, part, M, D, act_slope, rpw, nwg, st, dadd)
           : io_dtype == TSASR_BF16 ? launch_ln_bwd<bf16_t>(dy, x, gamma, beta, mean, rstd, dx, part, M, D, act_slope, rpw, nwg, st, dadd)
                                    : -2;
    TSASR_CHECK_ARG(rc == 0, "tsasr_layernorm_bwd: D=%d too large or bad io_dtype %d", D, io_dtype);
    if (tsasr_reduce_deferring()) {
        tsasr_reduce_submit(part, dgamma, 2 * D, nwg, D, 0, st);
        tsasr_reduce_submit(part + D, dbeta, 2 * D, nwg, D, 0, st);
    } else colsum_kernel<<<cdiv(2 * D, 16), 256, 0, st>>>(part, dgamma, dbeta, nwg, D, 2 * D);
    TSASR_CHECK_LAUNCH("tsasr_layernorm_bwd");
    return 0;
}

extern "C" {

int tsasr_layernorm_bwd(const void *dy, const void *x, const float *gamma, const float *beta, const float *mean,
                        const float *rstd, void *dx, float *dgamma, float *dbeta, long long M, int D, float act_slope,
                        int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    return layernorm_bwd_impl(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, M, D, act_slope, io_dtype, workspace, workspace_bytes, stream,
                              nullptr);
}

/* tsasr_layernorm_bwd with dx = LayerNorm_bwd(dy) + dadd: the gradient that reached x along a residual path is summed in the same
 * pass (a Conformer layer reads its input twice: through ffn_module1's LayerNorm and as the residual). Rows of at most 4 x 64 x 8
 * (bf16) / 4 x 64 x 4 (fp32) elements. */
int tsasr_layernorm_bwd_add(const void *dy, const void *dadd, const void *x, const float *gamma, const float *beta, const float *mean,
                            const float *rstd, void *dx, float *dgamma, float *dbeta, long long M, int D, float act_slope,
                            int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dadd, "tsasr_layernorm_bwd_add: null pointer");
    return layernorm_bwd_impl(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, M, D, act_slope, io_dtype, workspace, workspace_bytes, stream,
                              dadd);
}

static size_t slot_lds(int Ncols, int vec, const float *part) {
    const int chunks = Ncols / vec, tpr = chunks < 256 ? chunks : 256, slots = 256 / tpr;
    return (part && slots > 1) ? (size_t)slots * Ncols * sizeof(float) : 0;
}

static unsigned ew_grid(long long total, int N) {
    long long blocks = (total / N + 255) / 256;
    static const long long cap = 4096;
    if (blocks > cap) blocks = cap;  // grid-stride beyond 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int tsasr_bias_act_dropout_fwd(const void *x, const float *bias, void *y, long long M, int N, float act_slope, float p,
                               unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && y, "tsasr_bias_act_dropout_fwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && p >= 0.f && p < 1.f, "tsasr_bias_act_dropout_fwd: bad shape/p (M=%lld N=%d p=%f)", M, N, p);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32)
        bias_act_dropout_fwd_kernel<float><<<ew_grid(M * N, 4), 256, 0, st>>>((const float *)x, bias, (float *)y, M * N, N, act_slope, p, seed, seed_dev);
    else if (io_dtype == TSASR_BF16)
        bias_act_dropout_fwd_kernel<bf16_t><<<ew_grid(M * N, 8), 256, 0, st>>>((const bf16_t *)x, bias, (bf16_t *)y, M * N, N, act_slope, p, seed, seed_dev);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_bias_act_dropout_fwd");
    return 0;
}

size_t tsasr_colpart_workspace_bytes(long long M, int N) {
    const int rpw = pick_rows_per_wg(M, 16);
    return align_up((size_t)((M + rpw - 1) / rpw) * N * sizeof(float), 256);
}

int tsasr_bias_act_dropout_bwd(const void *dy, const void *y, void *dx, float *dbias, long long M, int N, float act_slope,
                               float p, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                               size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dy && dx && (act_slope < 0.f || y), "tsasr_bias_act_dropout_bwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0, "tsasr_bias_act_dropout_bwd: bad shape");
    TSASR_CHECK_ARG(!dbias || (workspace && workspace_bytes >= tsasr_colpart_workspace_bytes(M, N)), "tsasr_bias_act_dropout_bwd: workspace too small");
    const int rpw = pick_rows_per_wg(M, 16);
    const int nwg = (int)((M + rpw - 1) / rpw);
    float *part = dbias ? (float *)workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32)
        bias_act_dropout_bwd_kernel<float><<<nwg, 256, slot_lds(N, 4, part), st>>>((const float *)dy, (const float *)y, (float *)dx, part, M, N, act_slope, p, seed, seed_dev, rpw);
    else if (io_dtype == TSASR_BF16)
        bias_act_dropout_bwd_kernel<bf16_t><<<nwg, 256, slot_lds(N, 8, part), st>>>((const bf16_t *)dy, (const bf16_t *)y, (bf16_t *)dx, part, M, N, act_slope, p, seed, seed_dev, rpw);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    if (dbias) tsasr_reduce_submit(part, dbias, N, nwg, N, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_bias_act_dropout_bwd");
    return 0;
}

static int dropout_add_fwd_impl(const void *x, const float *bias, const void *res, void *out, long long M, int N, float alpha,
                                float p, unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows,
                                float p2, unsigned long long seed2, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && out, "tsasr_dropout_add_fwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && p >= 0.f && p < 1.f && p2 >= 0.f && p2 < 1.f, "tsasr_dropout_add_fwd: bad shape/p");
    TSASR_CHECK_ARG(!valid_lens || (Trows > 0 && M % Trows == 0), "tsasr_dropout_add_fwd: rows %lld not a multiple of T=%d", M, Trows);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32)
        dropout_add_fwd_kernel<float><<<ew_grid(M * N, 4), 256, 0, st>>>((const float *)x, bias, (const float *)res, (float *)out, M * N, N, alpha, p, seed, seed_dev, valid_lens, Trows, p2, seed2);
    else if (io_dtype == TSASR_BF16)
        dropout_add_fwd_kernel<bf16_t><<<ew_grid(M * N, 8), 256, 0, st>>>((const bf16_t *)x, bias, (const bf16_t *)res, (bf16_t *)out, M * N, N, alpha, p, seed, seed_dev, valid_lens, Trows, p2, seed2);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_dropout_add_fwd");
    return 0;
}

static int dropout_add_bwd_impl(const void *dout, void *dx, void *dres, float *dbias, long long M, int N, float alpha, float p,
                                unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows, float p2,
                                unsigned long long seed2, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dout && dx, "tsasr_dropout_add_bwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0, "tsasr_dropout_add_bwd: bad shape");
    TSASR_CHECK_ARG(!dbias || (workspace && workspace_bytes >= tsasr_colpart_workspace_bytes(M, N)), "tsasr_dropout_add_bwd: workspace too small");
    const int rpw = pick_rows_per_wg(M, 16);
    const int nwg = (int)((M + rpw - 1) / rpw);
    float *part = dbias ? (float *)workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32)
        dropout_add_bwd_kernel<float><<<nwg, 256, slot_lds(N, 4, part), st>>>((const float *)dout, (float *)dx, part, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, rpw, p2, seed2, (float *)dres);
    else if (io_dtype == TSASR_BF16)
        dropout_add_bwd_kernel<bf16_t><<<nwg, 256, slot_lds(N, 8, part), st>>>((const bf16_t *)dout, (bf16_t *)dx, part, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, rpw, p2, seed2, (bf16_t *)dres);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    if (dbias) tsasr_reduce_submit(part, dbias, N, nwg, N, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_dropout_add_bwd");
    return 0;
}

int tsasr_dropout_add_fwd(const void *x, const float *bias, const void *res, void *out, long long M, int N, float alpha,
                          float p, unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows,
                          int io_dtype, void *stream) {
    return dropout_add_fwd_impl(x, bias, res, out, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, 0.f, 0ull, io_dtype, stream);
}

int tsasr_dropout_add_bwd(const void *dout, void *dx, float *dbias, long long M, int N, float alpha, float p,
                          unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows, int io_dtype,
                          void *workspace, size_t workspace_bytes, void *stream) {
    return dropout_add_bwd_impl(dout, dx, nullptr, dbias, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, 0.f, 0ull, io_dtype, workspace,
                                workspace_bytes, stream);
}

/* out = dropout_p2( res + alpha * timemask(dropout_p(x + bias)) ): the residual tail of a front-end ConvBlock with its outer Dropout
 * (speechbrain/lobes/models/convolution.py:260-266: drop(drop(act(LN(conv(x)))) + LN(conv1x1(x)))) in one pass instead of two.
 * Backward: dres = dropout_p2'(dout) (written; it is also the gradient of `res`), dx = alpha * timemask * dropout_p'(dres). */
int tsasr_dropout_add2_fwd(const void *x, const float *bias, const void *res, void *out, long long M, int N, float alpha, float p,
                           unsigned long long seed, float p2, unsigned long long seed2, const unsigned long long *seed_dev,
                           const int32_t *valid_lens, int Trows, int io_dtype, void *stream) {
    return dropout_add_fwd_impl(x, bias, res, out, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, p2, seed2, io_dtype, stream);
}

int tsasr_dropout_add2_bwd(const void *dout, void *dx, void *dres, float *dbias, long long M, int N, float alpha, float p,
                           unsigned long long seed, float p2, unsigned long long seed2, const unsigned long long *seed_dev,
                           const int32_t *valid_lens, int Trows, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(p2 <= 0.f || dres, "tsasr_dropout_add2_bwd: dres is required when the outer dropout is active");
    return dropout_add_bwd_impl(dout, dx, dres, dbias, M, N, alpha, p, seed, seed_dev, valid_lens, Trows, p2, seed2, io_dtype, workspace,
                                workspace_bytes, stream);
}

/* s = res + alpha * timemask(dropout_p(x + bias)) ; y = LayerNorm(s; gamma, beta, eps).  x, res, s, y: [M, D] io_dtype
 * (D % 8 == 0, D <= 2048); mean, rstd fp32 [M] saved for the backward; valid_lens/Trows as tsasr_dropout_add_fwd. */
int tsasr_add_layernorm_fwd(const void *x, const float *bias, const void *res, void *s, void *y, float *mean, float *rstd,
                            const float *gamma, const float *beta, long long M, int D, float alpha, float p, unsigned long long seed,
                            const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows, float eps, int io_dtype,
                            void *stream) {
    TSASR_CHECK_ARG(x && res && s && y && mean && rstd && gamma && beta, "tsasr_add_layernorm_fwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && p >= 0.f && p < 1.f, "tsasr_add_layernorm_fwd: bad shape/p");
    TSASR_CHECK_ARG(!valid_lens || (Trows > 0 && M % Trows == 0), "tsasr_add_layernorm_fwd: rows not a multiple of T");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((M + 3) / 4);
#define ALN_F(TT, IT) add_layernorm_fwd_kernel<TT, IT><<<grid, 256, 0, st>>>((const TT *)x, bias, (const TT *)res, (TT *)s, (TT *)y, mean, rstd, gamma, beta, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, eps)
    static const int half_rows = 1;
    if (io_dtype == TSASR_BF16) {
        if (half_rows && D <= 256)
            add_layernorm_fwd_kernel<bf16_t, 1, true><<<(unsigned)((M + 7) / 8), 256, 0, st>>>((const bf16_t *)x, bias, (const bf16_t *)res, (bf16_t *)s, (bf16_t *)y, mean, rstd, gamma, beta, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, eps);
        else if (D <= 512) ALN_F(bf16_t, 1); else if (D <= 1024) ALN_F(bf16_t, 2); else if (D <= 2048) ALN_F(bf16_t, 4);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm_fwd: D=%d too large", D);
    } else if (io_dtype == TSASR_F32) {
        if (D <= 256) ALN_F(float, 1); else if (D <= 512) ALN_F(float, 2); else if (D <= 1024) ALN_F(float, 4); else if (D <= 2048) ALN_F(float, 8);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm_fwd: D=%d too large", D);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
#undef ALN_F
    TSASR_CHECK_LAUNCH("tsasr_add_layernorm_fwd");
    return 0;
}

size_t tsasr_add_layernorm_bwd_workspace_bytes(long long M, int D) {
    const int rpw = pick_rows_per_wg(M, 16);
    return align_up((size_t)((M + rpw - 1) / rpw) * 3 * D * sizeof(float), 256);
}

/* dres = LayerNorm_bwd(dy) + dout (dout may be NULL); dx = alpha * timemask * dropmask/(1-p) * dres; dgamma, dbeta, dbias fp32 [D]
 * (any of the three may be NULL). s = the tensor written by the forward. */
int tsasr_add_layernorm_bwd(const void *dy, const void *dout, const void *s, const float *gamma, const float *mean, const float *rstd,
                            void *dres, void *dx, float *dgamma, float *dbeta, float *dbias, long long M, int D, float alpha, float p,
                            unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows,
                            int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dy && s && gamma && mean && rstd && dres && dx && workspace, "tsasr_add_layernorm_bwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "tsasr_add_layernorm_bwd: bad shape");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_add_layernorm_bwd_workspace_bytes(M, D), "tsasr_add_layernorm_bwd: workspace too small");
    const int rpw = pick_rows_per_wg(M, 16);
    const int nwg = (int)((M + rpw - 1) / rpw);
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const size_t lds = (size_t)12 * D * sizeof(float);
#define ALN_B(TT, IT) add_layernorm_bwd_kernel<TT, IT><<<nwg, 256, lds, st>>>((const TT *)dy, (const TT *)dout, (const TT *)s, gamma, mean, rstd, (TT *)dres, (TT *)dx, part, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, rpw)
    static const int half_rows = 1;
    if (io_dtype == TSASR_BF16) {
        if (half_rows && D <= 256)
            add_layernorm_bwd_kernel<bf16_t, 1, true><<<nwg, 256, 2 * lds, st>>>((const bf16_t *)dy, (const bf16_t *)dout, (const bf16_t *)s, gamma, mean, rstd, (bf16_t *)dres, (bf16_t *)dx, part, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, rpw);
        else if (D <= 512) ALN_B(bf16_t, 1); else if (D <= 1024) ALN_B(bf16_t, 2); else if (D <= 2048) ALN_B(bf16_t, 4);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm_bwd: D=%d too large", D);
    } else if (io_dtype == TSASR_F32) {
        if (D <= 256) ALN_B(float, 1); else if (D <= 512) ALN_B(float, 2); else if (D <= 1024) ALN_B(float, 4); else if (D <= 2048) ALN_B(float, 8);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm_bwd: D=%d too large", D);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
#undef ALN_B
    if (tsasr_reduce_deferring()) {
        tsasr_reduce_submit(part, dgamma, 3 * D, nwg, D, 0, st);
        tsasr_reduce_submit(part + D, dbeta, 3 * D, nwg, D, 0, st);
        tsasr_reduce_submit(part + 2 * D, dbias, 3 * D, nwg, D, 0, st);
    } else colsum3_kernel<<<cdiv(3 * D, 16), 256, 0, st>>>(part, dgamma, dbeta, dbias, nwg, D);
    TSASR_CHECK_LAUNCH("tsasr_add_layernorm_bwd");
    return 0;
}

/* (s, y, z) = (res + alpha*timemask(dropout(x + bias)), LN(s; gamma, beta), LN(y; gamma2, beta2)): tsasr_add_layernorm_fwd followed by
 * tsasr_layernorm_fwd in one launch, bit-identical outputs (models/conformer.py:223-233 + Conformer.py:194-259: norm2 of a layer and
 * the next layer's first LayerNorm, or the encoder's final norm, whose eps differs: eps2). mean2 / rstd2 [M] are the statistics of y. */
int tsasr_add_layernorm2_fwd(const void *x, const float *bias, const void *res, void *s, void *y, void *z, float *mean, float *rstd,
                             float *mean2, float *rstd2, const float *gamma, const float *beta, const float *gamma2, const float *beta2,
                             long long M, int D, float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev,
                             const int32_t *valid_lens, int Trows, float eps, float eps2, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && res && s && y && z && mean && rstd && mean2 && rstd2 && gamma && beta && gamma2 && beta2, "tsasr_add_layernorm2_fwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && p >= 0.f && p < 1.f, "tsasr_add_layernorm2_fwd: bad shape/p");
    TSASR_CHECK_ARG(!valid_lens || (Trows > 0 && M % Trows == 0), "tsasr_add_layernorm2_fwd: rows not a multiple of T");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((M + 3) / 4);
#define ALN2_F(TT, IT, HWV, GR) add_layernorm2_fwd_kernel<TT, IT, HWV><<<GR, 256, 0, st>>>((const TT *)x, bias, (const TT *)res, (TT *)s, (TT *)y, (TT *)z, mean, rstd, mean2, rstd2, gamma, beta, gamma2, beta2, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, eps, eps2)
    if (io_dtype == TSASR_BF16) {
        if (D <= 256) ALN2_F(bf16_t, 1, true, (unsigned)((M + 7) / 8));
        else if (D <= 512) ALN2_F(bf16_t, 1, false, grid); else if (D <= 1024) ALN2_F(bf16_t, 2, false, grid);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm2_fwd: D=%d too large", D);
    } else if (io_dtype == TSASR_F32) {
        if (D <= 256) ALN2_F(float, 1, false, grid); else if (D <= 512) ALN2_F(float, 2, false, grid); else if (D <= 1024) ALN2_F(float, 4, false, grid);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm2_fwd: D=%d too large", D);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
#undef ALN2_F
    TSASR_CHECK_LAUNCH("tsasr_add_layernorm2_fwd");
    return 0;
}

size_t tsasr_add_layernorm2_bwd_workspace_bytes(long long M, int D) {
    const int rpw = pick_rows_per_wg(M, 16);
    return align_up((size_t)((M + rpw - 1) / rpw) * 5 * D * sizeof(float), 256);
}

/* Backward of tsasr_add_layernorm2_fwd: dz = gradient of z, dy = gradient reaching y along other paths (NULL: none), dout = gradient
 * reaching s along other paths (NULL: none). dres / dx as tsasr_add_layernorm_bwd; dgamma2 / dbeta2 belong to the second LayerNorm. */
int tsasr_add_layernorm2_bwd(const void *dz, const void *dy, const void *dout, const void *s, const float *gamma, const float *beta,
                             const float *gamma2, const float *mean, const float *rstd, const float *mean2, const float *rstd2, void *dres,
                             void *dx, float *dgamma, float *dbeta, float *dbias, float *dgamma2, float *dbeta2, long long M, int D,
                             float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens,
                             int Trows, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dz && s && gamma && beta && gamma2 && mean && rstd && mean2 && rstd2 && dres && dx && workspace, "tsasr_add_layernorm2_bwd: null pointer");
    TSASR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "tsasr_add_layernorm2_bwd: bad shape");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_add_layernorm2_bwd_workspace_bytes(M, D), "tsasr_add_layernorm2_bwd: workspace too small");
    const int rpw = pick_rows_per_wg(M, 16);
    const int nwg = (int)((M + rpw - 1) / rpw);
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
#define ALN2_B(TT, IT, HWV) add_layernorm2_bwd_kernel<TT, IT, HWV><<<nwg, 256, (size_t)(HWV ? 8 : 4) * 5 * D * sizeof(float), st>>>((const TT *)dz, (const TT *)dy, (const TT *)dout, (const TT *)s, gamma, beta, gamma2, mean, rstd, mean2, rstd2, (TT *)dres, (TT *)dx, part, M, D, alpha, p, seed, seed_dev, valid_lens, Trows, rpw)
    if (io_dtype == TSASR_BF16) {
        if (D <= 256) ALN2_B(bf16_t, 1, true); else if (D <= 512) ALN2_B(bf16_t, 1, false); else if (D <= 1024) ALN2_B(bf16_t, 2, false);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm2_bwd: D=%d too large", D);
    } else if (io_dtype == TSASR_F32) {
        if (D <= 256) ALN2_B(float, 1, false); else if (D <= 512) ALN2_B(float, 2, false); else if (D <= 1024) ALN2_B(float, 4, false);
        else TSASR_CHECK_ARG(false, "tsasr_add_layernorm2_bwd: D=%d too large", D);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
#undef ALN2_B
    tsasr_reduce_submit(part, dgamma, 5 * D, nwg, D, 0, st);
    tsasr_reduce_submit(part + D, dbeta, 5 * D, nwg, D, 0, st);
    tsasr_reduce_submit(part + 2 * D, dbias, 5 * D, nwg, D, 0, st);
    tsasr_reduce_submit(part + 3 * D, dgamma2, 5 * D, nwg, D, 0, st);
    tsasr_reduce_submit(part + 4 * D, dbeta2, 5 * D, nwg, D, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_add_layernorm2_bwd");
    return 0;
}

size_t tsasr_colsum_workspace_bytes(long long M, int N) {
    const long long rpw = std::max<long long>(64, (M + 1023) / 1024);
    return align_up((size_t)((M + rpw - 1) / rpw) * N * sizeof(float), 256);
}

/* out[c] (+)= sum_m x[m][c] for x [M, N] in io_dtype (N % 8 == 0, N <= 2048): the bias gradient of a layer whose output gradient
 * sits in HBM anyway (front-end convolutions: speechbrain/nnet/CNN.py:629-676 through the GEMM path). Partial rows per workgroup,
 * finished by the (deferrable) batched reduction. */
int tsasr_colsum(const void *x, float *out, long long M, int N, int accumulate, int io_dtype, void *workspace, size_t workspace_bytes,
                 void *stream) {
    TSASR_CHECK_ARG(x && out && workspace && M > 0 && N > 0 && N % 8 == 0 && N <= 2048, "tsasr_colsum: bad arguments");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_colsum_workspace_bytes(M, N), "tsasr_colsum: workspace too small");
    const int rpw = (int)std::max<long long>(64, (M + 1023) / 1024);
    const int nwg = (int)((M + rpw - 1) / rpw);
    float *part = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32) {
        const int slots = 256 / std::min(256, N / 4);
        colsum_rows_kernel<float><<<nwg, 256, (size_t)slots * N * sizeof(float), st>>>((const float *)x, part, M, N, rpw);
    } else if (io_dtype == TSASR_BF16) {
        const int slots = 256 / std::min(256, N / 8);
        colsum_rows_kernel<bf16_t><<<nwg, 256, (size_t)slots * N * sizeof(float), st>>>((const bf16_t *)x, part, M, N, rpw);
    } else TSASR_CHECK_ARG(false, "tsasr_colsum: bad io_dtype %d", io_dtype);
    tsasr_reduce_submit(part, out, N, nwg, N, accumulate, st);
    TSASR_CHECK_LAUNCH("tsasr_colsum");
    return 0;
}

}  // extern "C"
