// Small device-side pieces of the recipe / training loop that the reference runs as chains of tiny ATen kernels (each one a launch in
// the step's hipGraph): the masked mean-pool of the speaker encoder's output, the relative -> absolute length rounding, the
// non-finite-loss counter.
#include "common.h"

// ---- masked mean pool over time: out[b, d] = sum_{t < n_b} x[b, t, d] / n_b,  n_b = min(ceil(rel_b * T), T)
// (train_librispeechmix_scratch.py:52-64: length_to_mask(ceil(len * T).clamp(max=T)), masked sum / mask sum)
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const T *__restrict__ x, const float *__restrict__ rel, T *__restrict__ out, int Tn, int D) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int n = max(1, min((int)ceilf(rel[b] * (float)Tn), Tn));
    float s = 0.f;
    if (d < D)
        for (int t = sl; t < n; t += 4) s += ld1(x + ((long long)b * Tn + t) * D + d);
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && d < D) st1(out + (long long)b * D + d, ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)n);
}

template <typename T>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const T *__restrict__ dout, const float *__restrict__ rel, T *__restrict__ dx, int Tn, int D, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int d = (int)(e % D), t = (int)((e / D) % Tn), b = (int)(e / ((long long)D * Tn));
    const int n = max(1, min((int)ceilf(rel[b] * (float)Tn), Tn));
    st1(dx + e, t < n ? ld1(dout + (long long)b * D + d) / (float)n : 0.f);
}

// ---- out[k][b] = round-half-even(rel_k[b] * dim_k) (mode 0: models/conformer.py:272, SB/nnet/losses.py:58-59), floor (mode 1:
// SB/nnet/RNN.py:35 `.long()` of lengths * T) or ceil clamped to dim (mode 2): up to 8 length vectors in one launch
struct LenJobs { const float *rel[8]; int *out[8]; int dim[8], mode[8], n; };
__global__ void abs_lengths_kernel(LenJobs j, int B) {
    const int k = blockIdx.x, b = threadIdx.x;
    if (k >= j.n || b >= B) return;
    const float v = j.rel[k][b] * (float)j.dim[k];
    j.out[k][b] = j.mode[k] == 0 ? (int)rintf(v) : (j.mode[k] == 1 ? (int)floorf(v) : min((int)ceilf(v), j.dim[k]));
}

// ---- counter += number of non-finite values among x[0..n)  (SB/core.py:1115-1150 check_gradients counts non-finite losses)
__global__ void count_nonfinite_kernel(const float *__restrict__ x, int n, int *__restrict__ counter) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) c += !(fabsf(x[i]) <= 3.0e38f);
    c = (int)wave_sum((float)c);
    if (threadIdx.x == 0 && c) *counter += c;
}

// ---- speaker-embedding injection `sum` / `prod` (models/conformer.py:247-253): out[b,t,:] = src[b,t,:] (+ | *) spk[b,0,:]; backward in ONE
// pass over dout: dsrc = dout (sum) | dout * spk (prod), dspk[b,:] = sum_t dout (sum) | sum_t dout * src (prod), fixed summation order
template <typename T, bool PROD>
__global__ __launch_bounds__(256) void inject_fwd_kernel(const T *__restrict__ src, const T *__restrict__ spk, T *__restrict__ out, int Tn, int D, long long total) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (e >= total) return;
    const int d = (int)(e % D);
    const long long b = e / ((long long)D * Tn);
    float x[8], s[8];
    ld8(src + e, x);
    ld8(spk + b * D + d, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = PROD ? x[i] * s[i] : x[i] + s[i];
    st8(out + e, x);
}

// workgroup = (64 channels, utterance): 1024 threads = (64 / VE channel groups of one 16-byte piece) x NSL time slices; the slices' partial
// sums meet in LDS and are added in slice order (deterministic). (Round 4's form - 4 slices of 2-byte loads - took 263 us at B = 1, T' = 4000,
// where 4 workgroups walked 1000 frames each, and 18 us at the headline shape.)
template <typename T, bool PROD>
__global__ __launch_bounds__(1024) void inject_bwd_kernel(const T *__restrict__ dout, const T *__restrict__ src, const T *__restrict__ spk,
                                                          T *__restrict__ dsrc, T *__restrict__ dspk, int Tn, int D) {
    constexpr int VE = 16 / sizeof(T), NDL = 64 / VE, NSL = 1024 / NDL;
    __shared__ float red[NSL][64 + 1];
    const int b = blockIdx.y, dl = threadIdx.x % NDL, sl = threadIdx.x / NDL, d0 = blockIdx.x * 64 + dl * VE;
    const bool ok = d0 < D;         // D % 8 == 0 (checked by the launcher): a piece is inside or outside as a whole
    float sp[VE], acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sp[e] = (PROD && ok) ? ld1(spk + (long long)b * D + d0 + e) : 1.f; acc[e] = 0.f; }
    if (ok)
        for (int t = sl; t < Tn; t += NSL) {
            const long long e0 = ((long long)b * Tn + t) * D + d0;
            float g[VE], x[VE];
            if constexpr (VE == 8) ld8(dout + e0, g);
            else { const float4 v = *reinterpret_cast<const float4 *>(dout + e0); g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w; }
            if (PROD) {
                if constexpr (VE == 8) ld8(src + e0, x);
                else { const float4 v = *reinterpret_cast<const float4 *>(src + e0); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
                float o[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) { acc[e] += g[e] * x[e]; o[e] = g[e] * sp[e]; }
                if constexpr (VE == 8) st8(dsrc + e0, o);
                else *reinterpret_cast<float4 *>(dsrc + e0) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[e] += g[e];
                if (dsrc) {     // (sum: the gradient of src IS dout - the host passes NULL and hands dout on)
                    if constexpr (VE == 8) st8(dsrc + e0, g);
                    else *reinterpret_cast<float4 *>(dsrc + e0) = make_float4(g[0], g[1], g[2], g[3]);
                }
            }
        }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[sl][dl * VE + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < D) {
        float s = 0.f;
        for (int k = 0; k < NSL; ++k) s += red[k][threadIdx.x];
        st1(dspk + (long long)b * D + blockIdx.x * 64 + threadIdx.x, s);
    }
}

extern "C" {

/* out [B,T,D] = src [B,T,D] (+ | *) spk [B,1,D]  (mode 0 = sum, 1 = prod; D % 8 == 0) */
int tsasr_inject_fwd(const void *src, const void *spk, void *out, int B, int T, int D, int mode, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(src && spk && out && B > 0 && T > 0 && D > 0 && D % 8 == 0 && (mode == 0 || mode == 1), "tsasr_inject_fwd: bad arguments");
    const long long total = (long long)B * T * D;
    const unsigned grid = (unsigned)((total / 8 + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32) {
        if (mode) inject_fwd_kernel<float, true><<<grid, 256, 0, st>>>((const float *)src, (const float *)spk, (float *)out, T, D, total);
        else inject_fwd_kernel<float, false><<<grid, 256, 0, st>>>((const float *)src, (const float *)spk, (float *)out, T, D, total);
    } else {
        if (mode) inject_fwd_kernel<bf16_t, true><<<grid, 256, 0, st>>>((const bf16_t *)src, (const bf16_t *)spk, (bf16_t *)out, T, D, total);
        else inject_fwd_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t *)src, (const bf16_t *)spk, (bf16_t *)out, T, D, total);
    }
    TSASR_CHECK_LAUNCH("tsasr_inject_fwd");
    return 0;
}

/* dsrc [B,T,D], dspk [B,1,D] from dout [B,T,D] (src, spk read only in mode 1; mode 0: dsrc may be NULL - it equals dout) */
int tsasr_inject_bwd(const void *dout, const void *src, const void *spk, void *dsrc, void *dspk, int B, int T, int D, int mode, int io_dtype,
                     void *stream) {
    TSASR_CHECK_ARG(dout && dspk && (dsrc || mode == 0) && B > 0 && T > 0 && D > 0 && D % 8 == 0 && (mode == 0 || (mode == 1 && src && spk)), "tsasr_inject_bwd: bad arguments");
    dim3 grid(cdiv(D, 64), B);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32) {
        if (mode) inject_bwd_kernel<float, true><<<grid, 1024, 0, st>>>((const float *)dout, (const float *)src, (const float *)spk, (float *)dsrc, (float *)dspk, T, D);
        else inject_bwd_kernel<float, false><<<grid, 1024, 0, st>>>((const float *)dout, nullptr, nullptr, (float *)dsrc, (float *)dspk, T, D);
    } else {
        if (mode) inject_bwd_kernel<bf16_t, true><<<grid, 1024, 0, st>>>((const bf16_t *)dout, (const bf16_t *)src, (const bf16_t *)spk, (bf16_t *)dsrc, (bf16_t *)dspk, T, D);
        else inject_bwd_kernel<bf16_t, false><<<grid, 1024, 0, st>>>((const bf16_t *)dout, nullptr, nullptr, (bf16_t *)dsrc, (bf16_t *)dspk, T, D);
    }
    TSASR_CHECK_LAUNCH("tsasr_inject_bwd");
    return 0;
}

/* out[B,1,D] = masked mean over the first ceil(rel[b]*T) (clamped to T) frames of x [B,T,D] (io_dtype). */
int tsasr_mean_pool_fwd(const void *x, const float *rel, void *out, int B, int T, int D, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && rel && out && B > 0 && T > 0 && D > 0, "tsasr_mean_pool_fwd: bad arguments");
    dim3 grid(cdiv(D, 64), B);
    if (io_dtype == TSASR_F32) mean_pool_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float *)x, rel, (float *)out, T, D);
    else if (io_dtype == TSASR_BF16) mean_pool_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t *)x, rel, (bf16_t *)out, T, D);
    else TSASR_CHECK_ARG(false, "tsasr_mean_pool_fwd: bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_mean_pool_fwd");
    return 0;
}

/* dx[B,T,D] = dout[b,d] / n_b on the pooled frames, 0 elsewhere. */
int tsasr_mean_pool_bwd(const void *dout, const float *rel, void *dx, int B, int T, int D, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(dout && rel && dx && B > 0 && T > 0 && D > 0, "tsasr_mean_pool_bwd: bad arguments");
    const long long total = (long long)B * T * D;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (io_dtype == TSASR_F32) mean_pool_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float *)dout, rel, (float *)dx, T, D, total);
    else if (io_dtype == TSASR_BF16) mean_pool_bwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t *)dout, rel, (bf16_t *)dx, T, D, total);
    else TSASR_CHECK_ARG(false, "tsasr_mean_pool_bwd: bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_mean_pool_bwd");
    return 0;
}

/* Up to 8 absolute-length vectors in one launch: out[k][b] = f_k(rel[k][b] * dim[k]), B <= 1024; mode 0 = round half to even,
 * 1 = floor, 2 = ceil clamped to dim. rel / out: HOST arrays of n DEVICE pointers. */
int tsasr_abs_lengths(const float *const *rel, int *const *out, const int *dim, const int *mode, int n, int B, void *stream) {
    TSASR_CHECK_ARG(rel && out && dim && mode && n > 0 && n <= 8 && B > 0 && B <= 1024, "tsasr_abs_lengths: bad arguments (n=%d B=%d)", n, B);
    LenJobs j{};
    j.n = n;
    for (int k = 0; k < n; ++k) { j.rel[k] = rel[k]; j.out[k] = out[k]; j.dim[k] = dim[k]; j.mode[k] = mode[k]; }
    abs_lengths_kernel<<<n, (B + 63) / 64 * 64, 0, (hipStream_t)stream>>>(j, B);
    TSASR_CHECK_LAUNCH("tsasr_abs_lengths");
    return 0;
}

/* *counter += number of NaN / Inf among x[0..n) (fp32), one launch. */
int tsasr_count_nonfinite(const float *x, int n, int *counter, void *stream) {
    TSASR_CHECK_ARG(x && counter && n > 0, "tsasr_count_nonfinite: bad arguments");
    count_nonfinite_kernel<<<1, 64, 0, (hipStream_t)stream>>>(x, n, counter);
    TSASR_CHECK_LAUNCH("tsasr_count_nonfinite");
    return 0;
}

}  // extern "C"
