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

// TEST AID: LDS canary. A workgroup fills `words` 32-bit words of its own LDS with a pattern derived from its id, then verifies and
// rewrites them `iters` times; every word that does not read back as written is counted. Run beside another stream's kernels it
// detects writes that land outside their own LDS allocation (an LDS-DMA with a destination beyond the workgroup's allocation is not
// a `ds_write`: nothing says it is bounds-checked the same way).
__global__ __launch_bounds__(256) void lds_canary_kernel(int words, int iters, unsigned *__restrict__ errors, unsigned *__restrict__ first_bad) {
    extern __shared__ unsigned cw[];
    const unsigned key = 0xA5000000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += 256) cw[i] = key ^ (unsigned)i;
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = cw[i];
            if (v != (key ^ (unsigned)i)) {
                if (bad == 0 && first_bad) { first_bad[0] = blockIdx.x; first_bad[1] = (unsigned)i; first_bad[2] = v; first_bad[3] = key ^ (unsigned)i; }
                ++bad;
                cw[i] = key ^ (unsigned)i;
            }
        }
        __builtin_amdgcn_s_sleep(32);
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

// TEST AID: register canary. Every lane keeps 48 known values in VGPRs (pinned there by empty asm statements), sleeps, and checks them
// `iters` times: a value that changed was written by somebody else - this wave never writes them after the first assignment.
__global__ __launch_bounds__(256) void vgpr_canary_kernel(int iters, unsigned *__restrict__ errors, unsigned *__restrict__ first_bad) {
    unsigned v[48];
    const unsigned key = 0x5A000000u ^ ((blockIdx.x * 256u + threadIdx.x) * 2246822519u);
#pragma unroll
    for (int i = 0; i < 48; ++i) { v[i] = key + (unsigned)i * 0x01000193u; asm volatile("" : "+v"(v[i])); }
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            asm volatile("" : "+v"(v[i]));
            if (v[i] != key + (unsigned)i * 0x01000193u) {
                if (bad == 0 && first_bad) { first_bad[0] = blockIdx.x * 256u + threadIdx.x; first_bad[1] = (unsigned)i; first_bad[2] = v[i]; first_bad[3] = key + (unsigned)i * 0x01000193u; }
                ++bad;
                v[i] = key + (unsigned)i * 0x01000193u;
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
}

// TEST AID: barrier canary. The cross-wave pattern of the log-mel kernel (a table every wave fills a quarter of, one workgroup barrier,
// then every wave reads all of it): each round the four waves write round-dependent words, meet at the barrier, and every lane checks
// four words written by each OTHER wave. A mismatch = the barrier let a wave through before the others' LDS stores were visible.
__global__ __launch_bounds__(256) void barrier_canary_kernel(int rounds, unsigned *__restrict__ errors, unsigned *__restrict__ first_bad) {
    __shared__ unsigned pad0[4096];          // 16 KB in front (the log-mel kernel keeps its FFT buffers there)
    __shared__ unsigned tab[256 + 1280];     // the shared table sits at the log-mel kernel's offset of its twiddles (20496 B = 5124 words)
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    pad0[tid] = tid;
    unsigned bad = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const unsigned key = 0xC3000000u ^ (unsigned)(rd * 40503u) ^ (blockIdx.x << 8);
        tab[1028 + tid] = key + tid;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned idx = w * 64 + ((lane * 7 + rd) & 63), v = tab[1028 + idx];
            if (v != key + idx) {
                if (bad == 0 && first_bad) { first_bad[0] = blockIdx.x; first_bad[1] = (wave << 16) | (w << 8) | lane; first_bad[2] = v; first_bad[3] = key + idx; }
                ++bad;
            }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
    if (pad0[(tid * 5) & 4095] == 0xFFFFFFFFu) errors[1] = 1;   // keeps pad0 allocated
}

extern "C" {

/* TEST AID: `wgs` workgroups run `rounds` write / barrier / cross-wave read rounds; *errors (DEVICE uint[2]) += stale words seen. */
int tsasr_debug_barrier_canary(int wgs, int rounds, void *errors, void *first_bad, void *stream) {
    TSASR_CHECK_ARG(wgs > 0 && rounds > 0 && errors, "tsasr_debug_barrier_canary: bad arguments");
    barrier_canary_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(rounds, (unsigned *)errors, (unsigned *)first_bad);
    TSASR_CHECK_LAUNCH("tsasr_debug_barrier_canary");
    return 0;
}

/* TEST AID: `wgs` workgroups of 256 lanes keep 48 VGPRs each and verify them `iters` times; *errors (DEVICE uint) += changed registers. */
int tsasr_debug_vgpr_canary(int wgs, int iters, void *errors, void *first_bad, void *stream) {
    TSASR_CHECK_ARG(wgs > 0 && iters > 0 && errors, "tsasr_debug_vgpr_canary: bad arguments");
    vgpr_canary_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(iters, (unsigned *)errors, (unsigned *)first_bad);
    TSASR_CHECK_LAUNCH("tsasr_debug_vgpr_canary");
    return 0;
}

/* TEST AID: `wgs` workgroups hold `lds_bytes` of LDS each, check them `iters` times; *errors (device uint) += corrupted words seen,
 * first_bad (device uint[4], may be NULL) = {workgroup, word, value read, value expected} of one of them. */
int tsasr_debug_lds_canary(int wgs, int lds_bytes, int iters, void *errors, void *first_bad, void *stream) {
    TSASR_CHECK_ARG(wgs > 0 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && iters > 0 && errors, "tsasr_debug_lds_canary: bad arguments");
    (void)hipFuncSetAttribute((const void *)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    lds_canary_kernel<<<wgs, 256, lds_bytes, (hipStream_t)stream>>>(lds_bytes / 4, iters, (unsigned *)errors, (unsigned *)first_bad);
    TSASR_CHECK_LAUNCH("tsasr_debug_lds_canary");
    return 0;
}

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

/* TOOL AID: *out = the device's constant-rate wall clock (100 MHz) at the moment this one-thread kernel runs on `stream` - a captured
 * step carries a handful of these (prof.stamp, TSASR_STAMPS=1) to show when each phase REALLY starts in an unprofiled replay; rocprofv3's
 * kernel trace perturbs exactly that (tools/step_stamps.py). */
__global__ void stamp_kernel(unsigned long long *out) { *out = wall_clock64(); }
int tsasr_debug_stamp(void *out, void *stream) {
    TSASR_CHECK_ARG(out && ((uintptr_t)out & 7) == 0, "tsasr_debug_stamp: null or misaligned pointer");
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>((unsigned long long *)out);
    TSASR_CHECK_LAUNCH("tsasr_debug_stamp");
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
