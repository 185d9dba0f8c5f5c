// SpecAugment + speed perturbation: the two augmenters inside TSASR.compute_forward (TRAIN stage, `augment: True`).
//
//   reference: train_librispeechmix_scratch.py:82-94; speechbrain/lobes/augment.py:32-201 (SpecAugment: bicubic time warp,
//   frequency masks, time masks, fill = running global mean); speechbrain/processing/speech_augmentation.py:435-820
//   (SpeedPerturb = one of three polyphase windowed-sinc resamplers, Kaldi LinearResample).
//
// All of it is byte movement: [B,T,80] fp32 features (10 MB at B=32, T=1000) and [B,L] waveforms (20 MB) - HBM-bound, a few
// microseconds each at 8 TB/s; what the reference pays for is launches and host synchronisations (two interpolate calls with
// slice assignment, eight small tensor ops and a `.max()` host round trip per mask axis, 19-21 conv1d + conv_transpose1d + pad
// launches per resample). Here:
//   specaug_draw_kernel   one workgroup draws every random number of the call ON THE DEVICE (counter hash of the call seed + the
//                         device-resident step counter): no host round trip, and the call can sit inside a captured hipGraph;
//   specaug_warp_kernel   y = warp(x) in one pass (both halves, 4 clamped taps per output row), leaving per-workgroup partial
//                         sums of (everything, what the frequency masks cover, how many elements they cover);
//   specaug_mask_kernel   turns the partials into the two fill values (mean of the warped tensor; mean of the frequency-masked
//                         tensor = (S - S_masked + n_masked * mean1) / N, no second reduction pass) and applies both mask sets;
//   resample_kernel       one thread per output sample, the <= 21 x 25 filter bank in LDS.
#include <algorithm>

#include "common.h"

#define SA_MAX_MASKS 8     // masks per axis and utterance
#define SA_WG 256

// params (int32): [0] = c (warp centre; 0 = no warp), [1] = w (its new position), then flen[B*nf], fpos[B*nf], tlen[B*nt], tpos[B*nt]
__device__ __forceinline__ int sa_uniform(unsigned h, int lo, int hi) {   // integer in [lo, hi), hi > lo
    return lo + (int)(((unsigned long long)h * (unsigned)(hi - lo)) >> 32);
}

__global__ __launch_bounds__(SA_WG) void specaug_draw_kernel(int *__restrict__ params, int B, int T, int F, int window, int nf, int f_lo,
                                                              int f_hi, int nt, int t_lo, int t_hi, unsigned long long seed,
                                                              const unsigned long long *__restrict__ seed_dev) {
    __shared__ int red[SA_WG];
    const int tid = threadIdx.x;
    const DropKey key = drop_key(seed + (seed_dev ? *seed_dev : 0ull));
    if (tid == 0) {
        int c = 0, w = 0;
        if (window > 0 && T - window > window) {                       // augment.py:131-136
            c = sa_uniform(drop_hash(0, key), window, T - window);
            w = sa_uniform(drop_hash(1, key), c - window, c + window) + 1;
        }
        params[0] = c;
        params[1] = w;
    }
    int *flen = params + 2, *fpos = flen + B * nf, *tlen = fpos + B * nf, *tpos = tlen + B * nt;
    for (int axis = 0; axis < 2; ++axis) {
        const int n = axis == 0 ? B * nf : B * nt, lo = axis == 0 ? f_lo : t_lo, hi = axis == 0 ? f_hi : t_hi, D = axis == 0 ? F : T;
        int *len = axis == 0 ? flen : tlen, *pos = axis == 0 ? fpos : tpos;
        int mx = 0;
        for (int i = tid; i < n; i += SA_WG) {
            const int l = sa_uniform(drop_hash(16 + 4ull * i + 2 * axis, key), lo, hi);
            len[i] = l;
            mx = max(mx, l);
        }
        red[tid] = mx;
        __syncthreads();
        for (int s = SA_WG / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = max(red[tid], red[tid + s]);
            __syncthreads();
        }
        const int bound = max(1, D - red[0]);                          // ONE bound for the whole batch: D - mask_len.max() (:178-180)
        __syncthreads();
        for (int i = tid; i < n; i += SA_WG) pos[i] = sa_uniform(drop_hash(17 + 4ull * i + 2 * axis, key), 0, bound);
    }
}

// a * b rounded to fp32 BEFORE anything else uses it. The file is built with -ffp-contract=fast, under which the backend fuses ANY
// multiply into a following add (pragmas and HIP's __fmul_rn do not stop it): an fma of this product into `src - i0` moves the
// interpolation position by up to an ulp of src (1.5e-5 at row 200) away from the reference's. The empty asm hides the product.
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}

struct CubicTaps { float w0, w1, w2, w3; };
__device__ __forceinline__ CubicTaps cubic_taps(float t) {   // cubic convolution, A = -0.75 (what mode="bicubic" uses)
    const float A = -0.75f, u = 1.f - t;
    CubicTaps k;
    k.w0 = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    k.w1 = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    k.w2 = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
    k.w3 = ((A * (u + 1.f) - 5.f * A) * (u + 1.f) + 8.f * A) * (u + 1.f) - 4.f * A;
    return k;
}

__device__ __forceinline__ bool sa_masked(const int *len, const int *pos, int n, int i) {
    bool m = false;
    for (int k = 0; k < n; ++k) m |= (pos[k] <= i) & (i < pos[k] + len[k]);
    return m;
}

// One thread = one (b, t, 4 consecutive features). Row t of the output comes from the left part [0,c) -> [0,w) or the right part
// [c,T) -> [w,T), resized along time with align_corners=True (source position = i * (in-1)/(out-1)), taps clamped to their part.
template <typename T, int V>
__global__ __launch_bounds__(SA_WG) void specaug_warp_kernel(const T *__restrict__ x, T *__restrict__ y, const int *__restrict__ params,
                                                              int B, int Tn, int F, int nf, float *__restrict__ partial) {
    __shared__ float red[3][SA_WG / 64];
    const int c = params[0], w = params[1];
    const int *flen = params + 2, *fpos = flen + B * nf;
    const int FV = F / V;
    const long long total = (long long)B * Tn * FV;
    float s_all = 0.f, s_fm = 0.f, n_fm = 0.f;
    for (long long i = blockIdx.x * (long long)SA_WG + threadIdx.x; i < total; i += (long long)gridDim.x * SA_WG) {
        const int fv = (int)(i % FV), t = (int)((i / FV) % Tn), b = (int)(i / ((long long)FV * Tn));
        const T *xb = x + (long long)b * Tn * F + fv * V;
        float v[V];
        if (c > 0) {
            const bool left = t < w;
            const int base = left ? 0 : c, in_len = left ? c : Tn - c, out_len = left ? w : Tn - w, d = left ? t : t - w;
            const float scale = out_len > 1 ? (float)((double)(in_len - 1) / (double)(out_len - 1)) : 0.f;   // the device's fp32 divide is an ulp off the host's
            const float src = mul_rounded(scale, (float)d);
            const int i0 = min((int)floorf(src), in_len - 1);
            const CubicTaps k = cubic_taps(fminf(fmaxf(src - (float)i0, 0.f), 1.f));
            const int r0 = base + max(i0 - 1, 0), r1 = base + i0, r2 = base + min(i0 + 1, in_len - 1), r3 = base + min(i0 + 2, in_len - 1);
#pragma unroll
            for (int e = 0; e < V; ++e)
                v[e] = ((k.w0 * ld1(xb + (long long)r0 * F + e) + k.w1 * ld1(xb + (long long)r1 * F + e)) + k.w2 * ld1(xb + (long long)r2 * F + e)) +
                       k.w3 * ld1(xb + (long long)r3 * F + e);
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = ld1(xb + (long long)t * F + e);
        }
        T *yo = y + ((long long)b * Tn + t) * F + fv * V;
#pragma unroll
        for (int e = 0; e < V; ++e) st1(yo + e, v[e]);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float r = (float)(T)v[e];             // value as stored (bf16 io rounds here)
            s_all += r;
            if (nf > 0 && sa_masked(flen + b * nf, fpos + b * nf, nf, fv * V + e)) { s_fm += r; n_fm += 1.f; }
        }
    }
    s_all = wave_sum(s_all); s_fm = wave_sum(s_fm); n_fm = wave_sum(n_fm);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s_all; red[1][wave] = s_fm; red[2][wave] = n_fm; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int k = 0; k < SA_WG / 64; ++k) s += red[threadIdx.x][k];
        partial[blockIdx.x * 3 + threadIdx.x] = s;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(SA_WG) void specaug_mask_kernel(T *__restrict__ y, const int *__restrict__ params, int B, int Tn, int F, int nf,
                                                              int nt, int replace_with_zero, const float *__restrict__ partial, int nparts) {
    __shared__ double red[3][SA_WG];
    __shared__ float fill[2];
    {   // every workgroup folds the <= 1024 partial triples itself (12 KB, L2-resident) in a fixed order: no extra launch
        double a = 0, f = 0, n = 0;
        for (int p = threadIdx.x; p < nparts; p += SA_WG) { a += partial[p * 3]; f += partial[p * 3 + 1]; n += partial[p * 3 + 2]; }
        red[0][threadIdx.x] = a; red[1][threadIdx.x] = f; red[2][threadIdx.x] = n;
        __syncthreads();
        for (int s = SA_WG / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s)
                for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double N = (double)B * Tn * F;
            const double mean1 = red[0][0] / N;                                    // x.mean() before the frequency masks
            const double fill1 = replace_with_zero ? 0.0 : (double)(float)(T)(float)mean1;
            const double sum2 = nf > 0 ? red[0][0] - red[1][0] + red[2][0] * fill1 : red[0][0];
            fill[0] = replace_with_zero ? 0.f : (float)mean1;
            fill[1] = replace_with_zero ? 0.f : (float)(sum2 / N);                 // x.mean() after them, before the time masks
        }
        __syncthreads();
    }
    const int *flen = params + 2, *fpos = flen + B * nf, *tlen = fpos + B * nf, *tpos = tlen + B * nt;
    const int FV = F / V;
    const long long total = (long long)B * Tn * FV;
    const float v1 = fill[0], v2 = fill[1];
    for (long long i = blockIdx.x * (long long)SA_WG + threadIdx.x; i < total; i += (long long)gridDim.x * SA_WG) {
        const int fv = (int)(i % FV), t = (int)((i / FV) % Tn), b = (int)(i / ((long long)FV * Tn));
        T *yo = y + ((long long)b * Tn + t) * F + fv * V;
        const bool tm = nt > 0 && sa_masked(tlen + b * nt, tpos + b * nt, nt, t);
        if (tm) {
#pragma unroll
            for (int e = 0; e < V; ++e) st1(yo + e, v2);
        } else if (nf > 0) {
#pragma unroll
            for (int e = 0; e < V; ++e)
                if (sa_masked(flen + b * nf, fpos + b * nf, nf, fv * V + e)) st1(yo + e, v1);
        }
    }
}

// Output sample n = q * P + i of one utterance: sum_j weights[i][j] * x[first[i] + q * stride + j], zeros outside [0, L).
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ weights,
                                                        const int *__restrict__ first, int L, int n_out, int P, int stride, int W) {
    extern __shared__ float lds[];            // [P*W] weights, then [P] first indices
    int *fi = reinterpret_cast<int *>(lds + P * W);
    for (int i = threadIdx.x; i < P * W; i += 256) lds[i] = weights[i];
    for (int i = threadIdx.x; i < P; i += 256) fi[i] = first[i];
    __syncthreads();
    const int b = blockIdx.y;
    const float *xb = x + (long long)b * L;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < n_out; n += gridDim.x * 256) {
        const int i = n % P, q = n / P, start = fi[i] + q * stride;
        const float *wr = lds + i * W;
        float acc = 0.f;
        for (int j = 0; j < W; ++j) {
            const int p = start + j;
            const float v = xb[min(max(p, 0), L - 1)];          // always-issued clamped load, masked afterwards
            acc += (p >= 0 && p < L) ? v * wr[j] : 0.f;
        }
        y[(long long)b * n_out + n] = acc;
    }
}

static long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }

extern "C" {

/* int32 words of the SpecAugment draw table for a batch of B utterances (c, w, then widths and starts of every mask). */
size_t tsasr_specaug_params_words(int B, int n_freq_mask, int n_time_mask) { return 2 + (size_t)2 * B * (n_freq_mask + n_time_mask); }

/* Draw the random numbers of one SpecAugment call into `params` (device, tsasr_specaug_params_words int32):
 * c ~ U[window, T-window), w ~ U[c-window, c+window) + 1 (both 0 when window == 0 or T - window <= window); per utterance and mask
 * width ~ U[lo, hi) and start ~ U[0, max(1, D - max width over the batch)) - speechbrain/lobes/augment.py:131-136,173-180.
 * seed + *seed_dev selects the stream (seed_dev may be NULL); nothing is read back by the host. */
int tsasr_specaug_draw(int *params, int B, int T, int F, int window, int n_freq_mask, int f_lo, int f_hi, int n_time_mask, int t_lo,
                       int t_hi, unsigned long long seed, const unsigned long long *seed_dev, void *stream) {
    TSASR_CHECK_ARG(params && B > 0 && T > 0 && F > 0 && window >= 0, "tsasr_specaug_draw: bad arguments");
    TSASR_CHECK_ARG(n_freq_mask >= 0 && n_freq_mask <= SA_MAX_MASKS && n_time_mask >= 0 && n_time_mask <= SA_MAX_MASKS,
                    "tsasr_specaug_draw: at most %d masks per axis", SA_MAX_MASKS);
    TSASR_CHECK_ARG((n_freq_mask == 0 || (f_lo >= 0 && f_hi > f_lo)) && (n_time_mask == 0 || (t_lo >= 0 && t_hi > t_lo)),
                    "tsasr_specaug_draw: mask width ranges must be non-empty (freq [%d,%d), time [%d,%d))", f_lo, f_hi, t_lo, t_hi);
    specaug_draw_kernel<<<1, SA_WG, 0, (hipStream_t)stream>>>(params, B, T, F, window, n_freq_mask, f_lo, f_hi, n_time_mask, t_lo, t_hi,
                                                             seed, seed_dev);
    TSASR_CHECK_LAUNCH("tsasr_specaug_draw");
    return 0;
}

size_t tsasr_specaug_workspace_bytes(void) { return 1024 * 3 * sizeof(float); }

/* y = SpecAugment(x) for x, y [B,T,F] (io_dtype; y must not alias x) with the draws in `params` (device; layout of
 * tsasr_specaug_draw, params[0] == 0 = no time warp). Fill value of the masks: 0 if replace_with_zero, else the mean of the
 * whole tensor at that point of the pipeline (warp -> frequency masks -> time masks), as the reference computes it. */
int tsasr_specaug_apply(const void *x, void *y, const int *params, int B, int T, int F, int n_freq_mask, int n_time_mask,
                        int replace_with_zero, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(x && y && x != y && params && B > 0 && T > 0 && F > 0, "tsasr_specaug_apply: bad arguments");
    TSASR_CHECK_ARG(n_freq_mask >= 0 && n_freq_mask <= SA_MAX_MASKS && n_time_mask >= 0 && n_time_mask <= SA_MAX_MASKS,
                    "tsasr_specaug_apply: at most %d masks per axis", SA_MAX_MASKS);
    TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_specaug_workspace_bytes(), "tsasr_specaug_apply: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace;
    const int V = F % 4 == 0 ? 4 : 1;
    const long long total = (long long)B * T * (F / V);
    const int grid = (int)std::min<long long>(1024, (total + SA_WG - 1) / SA_WG);
#define SA_LAUNCH(TT, VV)                                                                                                              \
    do {                                                                                                                               \
        specaug_warp_kernel<TT, VV><<<grid, SA_WG, 0, st>>>((const TT *)x, (TT *)y, params, B, T, F, n_freq_mask, partial);            \
        if (n_freq_mask > 0 || n_time_mask > 0)                                                                                        \
            specaug_mask_kernel<TT, VV><<<grid, SA_WG, 0, st>>>((TT *)y, params, B, T, F, n_freq_mask, n_time_mask, replace_with_zero, \
                                                                 partial, grid);                                                      \
    } while (0)
    if (io_dtype == TSASR_F32) { if (V == 4) SA_LAUNCH(float, 4); else SA_LAUNCH(float, 1); }
    else if (io_dtype == TSASR_BF16) { if (V == 4) SA_LAUNCH(bf16_t, 4); else SA_LAUNCH(bf16_t, 1); }
    else TSASR_CHECK_ARG(false, "tsasr_specaug_apply: bad io_dtype %d", io_dtype);
#undef SA_LAUNCH
    TSASR_CHECK_LAUNCH("tsasr_specaug_apply");
    return 0;
}

/* Samples produced from n_in input samples when going from orig_freq to new_freq: the output instants k / new_freq inside
 * [0, n_in / orig_freq) - speech_augmentation.py:705-756. */
long long tsasr_resample_out_len(long long n_in, int orig_freq, int new_freq) {
    if (n_in <= 0 || orig_freq <= 0 || new_freq <= 0) return 0;
    const long long tick = (long long)orig_freq * new_freq / gcd_ll(orig_freq, new_freq);
    const long long span = n_in * (tick / orig_freq), per_out = tick / new_freq;
    long long last = span / per_out;
    if (last * per_out == span) --last;
    return last + 1;
}

/* Polyphase resampling of x [B,L] fp32 into y [B,n_out] (n_out = tsasr_resample_out_len). weights [P,W] fp32 and first [P] int32
 * (device): the filter bank and the first input index of each phase (P = new/gcd, stride = orig/gcd input samples per unit). */
int tsasr_resample_fwd(const float *x, float *y, const float *weights, const int *first, int B, int L, int n_out, int P, int stride,
                       int W, void *stream) {
    TSASR_CHECK_ARG(x && y && weights && first && B > 0 && L > 0 && n_out > 0 && P > 0 && stride > 0 && W > 0, "tsasr_resample_fwd: bad arguments");
    const size_t lds = ((size_t)P * W + P) * 4;
    TSASR_CHECK_ARG(lds <= 48 * 1024, "tsasr_resample_fwd: filter bank of %d x %d does not fit LDS", P, W);
    TSASR_CHECK_ARG(B <= 65535, "tsasr_resample_fwd: B too large");
    dim3 grid(std::min(4096, cdiv(n_out, 256)), B);
    resample_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(x, y, weights, first, L, n_out, P, stride, W);
    TSASR_CHECK_LAUNCH("tsasr_resample_fwd");
    return 0;
}

}  // extern "C"
