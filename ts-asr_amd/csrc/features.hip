// Log-mel filterbank front-end (A1) and per-utterance normalisation (A2) for gfx950.
//
// Replaces speechbrain/lobes/features.py:130-147 (Fbank.forward) = STFT speechbrain/processing/features.py:134-178
// (torch.stft: n_fft 512, hop 160, periodic Hamming window 512, center=True with zero padding, one-sided),
// spectral_magnitude power=1 (:317-348), Filterbank.forward (:482-552: triangular mel matrix, rebuilt on the CPU and
// copied to the device on EVERY call in the reference), _amplitude_to_DB (:683-704: 10 log10(max(x, 1e-10)), then a floor
// at (max over the whole padded utterance) - 80 dB); and InputNormalization(norm_type="sentence") (:1012-1025,1107-1132:
// a Python loop over the batch with one host sync per utterance).
//
// fbank: one WAVE per frame: 512 windowed samples -> in-LDS radix-2 FFT (bit-reversed load, 9 stages, 4 butterflies per
// lane per stage, twiddles from a 256-entry table in LDS) -> power spectrum -> 80 mel filters (lane = filter, dense fp32
// matrix read through L1/L2) -> dB, plus an order-preserving integer atomicMax per utterance; a second tiny kernel applies
// the -80 dB floor. Roughly 53 kFLOP and ~1 KB per frame => HBM/latency-bound; everything between the waveform read and
// the [T,80] write stays in LDS/registers (the reference makes 5 full passes through HBM).
// sentence_norm: one workgroup per utterance, two passes for mean and unbiased std over the valid frames (lanes own
// feature bins), one pass to write (x - mean)/max(std, eps) for ALL frames (padded ones too, as the reference does).
#include "common.h"

#define FB_N 512
#define FB_LOG2N 9

__device__ __forceinline__ unsigned bitrev9(unsigned x) { return __brev(x) >> (32 - FB_LOG2N); }
// order-preserving float <-> int map for atomicMax
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void fbank_kernel(const float *__restrict__ wav, const float *__restrict__ window,
                                                    const float *__restrict__ melmat /*[257][n_mels]*/, float *__restrict__ out_db,
                                                    int *__restrict__ umax, const int *__restrict__ band, int L, int Tn, int n_mels,
                                                    int hop, float amin) {
    __shared__ float2 buf[4][FB_N];
    __shared__ float2 tw[FB_N / 2];
    __shared__ float pw[4][FB_N / 2 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < FB_N / 2; i += 256) {
        float s, c;
        sincospif(-2.f * i / FB_N, &s, &c);
        tw[i] = make_float2(c, s);
    }
    const long long frame = (long long)blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    const bool live = frame < Tn;
    float2 *x = buf[wave];
    if (live) {
        const long long start = frame * hop - FB_N / 2;  // center=True: frame t covers [t*hop - 256, t*hop + 256)
        for (int i = lane; i < FB_N; i += 64) {
            const long long p = start + i;
            const float v = (p >= 0 && p < L) ? wav[(long long)b * L + p] * window[i] : 0.f;
            x[bitrev9(i)] = make_float2(v, 0.f);
        }
    }
    __syncthreads();
    for (int st = 0; st < FB_LOG2N; ++st) {
        const int half = 1 << st;
        if (live) {
            for (int k = lane; k < FB_N / 2; k += 64) {
                const int j = k & (half - 1), i0 = ((k >> st) << (st + 1)) + j, i1 = i0 + half;
                const float2 w = tw[j << (FB_LOG2N - 1 - st)];
                const float2 a = x[i0], c = x[i1];
                const float2 t = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
                x[i0] = make_float2(a.x + t.x, a.y + t.y);
                x[i1] = make_float2(a.x - t.x, a.y - t.y);
            }
        }
        __syncthreads();
    }
    if (live)
        for (int k = lane; k <= FB_N / 2; k += 64) pw[wave][k] = x[k].x * x[k].x + x[k].y * x[k].y;
    __syncthreads();
    float mx = -INFINITY;
    if (live) {
        for (int m = lane; m < n_mels; m += 64) {
            float s = 0.f;
            // triangular filters: bin m is non-zero on [band[m], band[n_mels + m]) only (fbank_bands_kernel) - ~6 of the 257
            // frequency rows instead of all of them, summed in the same ascending order as the dense loop
            const int k_lo = band[m], k_hi = band[n_mels + m];
            for (int k = k_lo; k < k_hi; ++k) {
                const float w = melmat[k * n_mels + m];
                if (w != 0.f) s += w * pw[wave][k];
            }
            const float db = 10.f * (float)log10((double)fmaxf(s, amin));  // fp64 log: -100 dB exactly for silence, as the reference test asserts
            out_db[((long long)b * Tn + frame) * n_mels + m] = db;
            mx = fmaxf(mx, db);
        }
    }
    mx = wave_max(mx);
    if (live && lane == 0) atomicMax(umax + b, f2ord(mx));
}

template <typename T>
__global__ __launch_bounds__(256) void fbank_floor_kernel(const float *__restrict__ db, const int *__restrict__ umax, T *__restrict__ out,
                                                          long long per_utt, float top_db) {
    const int b = blockIdx.y;
    const float floor_v = ord2f(umax[b]) - top_db;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_utt; i += (long long)gridDim.x * 256)
        st1(out + (long long)b * per_utt + i, fmaxf(db[(long long)b * per_utt + i], floor_v));
}

__global__ void fbank_init_kernel(int *umax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) umax[i] = f2ord(-INFINITY);
}

// band[m] = first non-zero frequency row of mel column m, band[n_mels + m] = one past the last (0, 0 for an all-zero column)
__global__ void fbank_bands_kernel(const float *__restrict__ melmat, int *__restrict__ band, int n_mels) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_mels) return;
    int lo = FB_N / 2 + 1, hi = 0;
    for (int k = 0; k <= FB_N / 2; ++k)
        if (melmat[k * n_mels + m] != 0.f) { lo = min(lo, k); hi = k + 1; }
    band[m] = hi > 0 ? lo : 0;
    band[n_mels + m] = hi;
}

#define SN_NT 1024   // one workgroup (16 waves) streams an utterance: rows of Fq bins are spread over SN_NT / Fq thread rows
template <typename TI, typename TO>
__global__ __launch_bounds__(SN_NT) void sentence_norm_kernel(const TI *__restrict__ x, const int32_t *__restrict__ lens, TO *__restrict__ y,
                                                              int Tn, int Fq, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [rows][Fq] partials, then mean[Fq], inv[Fq]
    const int b = blockIdx.x;
    const int n = min(max(lens[b], 1), Tn);
    const int f = threadIdx.x % Fq, rl = threadIdx.x / Fq, rows = SN_NT / Fq;
    const TI *xb = x + (long long)b * Tn * Fq;
    float *mean = sm + rows * Fq, *inv = mean + Fq;
    // column sums with four independent accumulators (loads of four frames in flight per thread)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (rl < rows) {
        int t = rl;
        for (; t + 3 * rows < n; t += 4 * rows) {
            s0 += ld1(xb + (long long)t * Fq + f);
            s1 += ld1(xb + (long long)(t + rows) * Fq + f);
            s2 += ld1(xb + (long long)(t + 2 * rows) * Fq + f);
            s3 += ld1(xb + (long long)(t + 3 * rows) * Fq + f);
        }
        for (; t < n; t += rows) s0 += ld1(xb + (long long)t * Fq + f);
        sm[rl * Fq + f] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (threadIdx.x < Fq) {
        float a = 0.f;
        for (int q = 0; q < rows; ++q) a += sm[q * Fq + threadIdx.x];
        mean[threadIdx.x] = a / n;
    }
    __syncthreads();
    const float mu = mean[f];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (rl < rows) {
        int t = rl;
        for (; t + 3 * rows < n; t += 4 * rows) {
            const float d0 = ld1(xb + (long long)t * Fq + f) - mu, d1 = ld1(xb + (long long)(t + rows) * Fq + f) - mu;
            const float d2 = ld1(xb + (long long)(t + 2 * rows) * Fq + f) - mu, d3 = ld1(xb + (long long)(t + 3 * rows) * Fq + f) - mu;
            v0 += d0 * d0; v1 += d1 * d1; v2 += d2 * d2; v3 += d3 * d3;
        }
        for (; t < n; t += rows) { const float d = ld1(xb + (long long)t * Fq + f) - mu; v0 += d * d; }
    }
    __syncthreads();   // pass-1 partials fully consumed
    if (rl < rows) sm[rl * Fq + f] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (threadIdx.x < Fq) {
        float a = 0.f;
        for (int q = 0; q < rows; ++q) a += sm[q * Fq + threadIdx.x];
        inv[threadIdx.x] = 1.f / fmaxf(sqrtf(a / (n - 1)), eps);  // unbiased; n == 1 gives NaN exactly as torch.std does
    }
    __syncthreads();
    TO *yb = y + (long long)b * Tn * Fq;
    if (rl < rows) {   // thread (rl, f) keeps its bin: no modulo in the loop
        const float m_f = mean[f], i_f = inv[f];
#pragma unroll 4
        for (int t = rl; t < Tn; t += rows) st1(yb + (long long)t * Fq + f, (ld1(xb + (long long)t * Fq + f) - m_f) * i_f);
    }
}

extern "C" {

size_t tsasr_fbank_workspace_bytes(int B, int T, int n_mels) {
    return align_up((size_t)B * T * n_mels * sizeof(float), 256) + align_up((size_t)B * sizeof(int), 256) + align_up((size_t)2 * n_mels * sizeof(int), 256);
}

/* wav [B,L] fp32 -> out [B, T = 1 + L/hop, n_mels] (out_dtype): log-mel in dB with the per-utterance (max - top_db) floor.
 * window [512] fp32 (Hamming, periodic), melmat [257, n_mels] fp32 (triangular filters). n_fft is fixed at 512. */
int tsasr_fbank_fwd(const float *wav, const float *window, const float *melmat, void *out, int B, int L, int T, int n_mels, int hop,
                    float top_db, float amin, int out_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(wav && window && melmat && out && workspace, "tsasr_fbank_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && L > 0 && hop > 0 && T == 1 + L / hop && n_mels > 0 && n_mels <= 128, "tsasr_fbank_fwd: bad shape (L=%d hop=%d T=%d n_mels=%d)", L, hop, T, n_mels);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_fbank_workspace_bytes(B, T, n_mels), "tsasr_fbank_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *db = (float *)workspace;
    int *umax = (int *)((char *)workspace + align_up((size_t)B * T * n_mels * sizeof(float), 256));
    int *band = (int *)((char *)umax + align_up((size_t)B * sizeof(int), 256));
    fbank_init_kernel<<<cdiv(B, 64), 64, 0, st>>>(umax, B);
    fbank_bands_kernel<<<cdiv(n_mels, 64), 64, 0, st>>>(melmat, band, n_mels);
    fbank_kernel<<<dim3(cdiv(T, 4), B), 256, 0, st>>>(wav, window, melmat, db, umax, band, L, T, n_mels, hop, amin);
    const long long per = (long long)T * n_mels;
    dim3 g2((unsigned)min((long long)64, (per + 255) / 256), B);
    if (out_dtype == TSASR_F32) fbank_floor_kernel<float><<<g2, 256, 0, st>>>(db, umax, (float *)out, per, top_db);
    else if (out_dtype == TSASR_BF16) fbank_floor_kernel<bf16_t><<<g2, 256, 0, st>>>(db, umax, (bf16_t *)out, per, top_db);
    else TSASR_CHECK_ARG(false, "tsasr_fbank_fwd: bad out_dtype %d", out_dtype);
    TSASR_CHECK_LAUNCH("tsasr_fbank_fwd");
    return 0;
}

/* y = (x - mean_b) / max(std_b, eps), statistics per utterance and feature bin over the first lens[b] frames (unbiased std). */
int tsasr_sentence_norm_fwd(const void *x, const int32_t *lens, void *y, int B, int T, int F, float eps, int in_dtype, int out_dtype,
                            void *stream) {
    TSASR_CHECK_ARG(x && lens && y, "tsasr_sentence_norm_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && F > 0 && F <= 256, "tsasr_sentence_norm_fwd: bad shape (F=%d must be <= 256)", F);
    const size_t lds = (size_t)((SN_NT / F) * F + 2 * F) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == TSASR_F32 && out_dtype == TSASR_F32) sentence_norm_kernel<float, float><<<B, SN_NT, lds, st>>>((const float *)x, lens, (float *)y, T, F, eps);
    else if (in_dtype == TSASR_F32 && out_dtype == TSASR_BF16) sentence_norm_kernel<float, bf16_t><<<B, SN_NT, lds, st>>>((const float *)x, lens, (bf16_t *)y, T, F, eps);
    else if (in_dtype == TSASR_BF16 && out_dtype == TSASR_BF16) sentence_norm_kernel<bf16_t, bf16_t><<<B, SN_NT, lds, st>>>((const bf16_t *)x, lens, (bf16_t *)y, T, F, eps);
    else if (in_dtype == TSASR_BF16 && out_dtype == TSASR_F32) sentence_norm_kernel<bf16_t, float><<<B, SN_NT, lds, st>>>((const bf16_t *)x, lens, (float *)y, T, F, eps);
    else TSASR_CHECK_ARG(false, "tsasr_sentence_norm_fwd: bad dtypes %d %d", in_dtype, out_dtype);
    TSASR_CHECK_LAUNCH("tsasr_sentence_norm_fwd");
    return 0;
}

}  // extern "C"
