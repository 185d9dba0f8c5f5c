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

// test aid: every CU's whole LDS is overwritten with `pattern` (a kernel that reads LDS it never wrote then sees this, not leftovers)
__global__ __launch_bounds__(256) void fill_lds_kernel(unsigned pattern, int words, unsigned *sink) {
    extern __shared__ unsigned fill_lds[];
    for (int i = threadIdx.x; i < words; i += 256) fill_lds[i] = pattern;
    __syncthreads();
    if (sink && fill_lds[(threadIdx.x * 97) % words] != pattern) *sink = 1;   // keeps the stores alive
}

__global__ void fill_words_kernel(unsigned *p, unsigned pattern, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pattern;
}

extern "C" {

/* TEST AID: fill the LDS of every CU with a 32-bit pattern (160 KB workgroups, enough of them that every CU runs at least one). */
int tsasr_debug_fill_lds(unsigned pattern, void *stream) {
    const int bytes = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)fill_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    fill_lds_kernel<<<4 * cus, 256, bytes, (hipStream_t)stream>>>(pattern, bytes / 4, nullptr);
    TSASR_CHECK_LAUNCH("tsasr_debug_fill_lds");
    return 0;
}

/* TEST AID: fill `nwords` 32-bit words of device memory with a pattern (tools/det_stress.py --poison: NaN into every inactive block of the
 * captured step's memory pool between replays - a kernel that reads a buffer before its producer of THIS replay wrote it then shows up as
 * NaN instead of as the previous replay's nearly identical values). */
int tsasr_debug_fill(void *p, unsigned pattern, size_t nwords, void *stream) {
    TSASR_CHECK_ARG(p && ((uintptr_t)p & 3) == 0, "tsasr_debug_fill: null or misaligned pointer");
    if (nwords == 0) return 0;
    fill_words_kernel<<<(unsigned)std::min<size_t>(4096, (nwords + 255) / 256), 256, 0, (hipStream_t)stream>>>((unsigned *)p, pattern, nwords);
    TSASR_CHECK_LAUNCH("tsasr_debug_fill");
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
