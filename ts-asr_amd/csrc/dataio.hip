// LibriSpeechMix mixture construction on the device (data side of the training step).
//
// Replaces the arithmetic of the reference's `audio_pipeline` (train_librispeechmix_scratch.py:356-386), which runs per utterance in
// DataLoader worker processes on the host:
//     gain_j  = sqrt(10^(g/10) * mean(target^2) / mean(src_j^2))   for every non-target source when gain_nontarget != 0   (:361-369)
//     src_j   = pad(src_j * gain_j, [ceil(delay_j * sr), 0])                                                                 (:370)
//     mixed   = src_0 + src_1 + ... (left to right, zero-padded to the longest)                                             (:372-377)
//     mixed   = mixed[ceil(start * sr) : ceil(start * sr) + ceil(duration * sr)]                                            (:379-385)
// Design: HBM-bound element-wise work. Two launches per utterance: (1) one workgroup per source sums its squares in fp64 in a
// fixed order (bit-reproducible; the correctly rounded mean power, where the reference's fp32 cascade sum depends on the host's SIMD
// width) and thread 0 of the LAST... no cross-workgroup step: gains are formed by the mix kernel itself from the nsrc power words;
// (2) the mix kernel walks the output window with 16-byte stores, adding the sources in index order in fp32 exactly as the reference's
// `mixed_sig += sig` loop does (a source that does not cover a sample contributes the +0.0 of its padding).
#include "common.h"

#define MIX_MAX_SRC 8

struct MixArgs {
    const float *src;                  // all sources of the utterance back to back
    long long off[MIX_MAX_SRC + 1];    // source j = src[off[j] .. off[j+1])
    int delay[MIX_MAX_SRC];            // ceil(delay_j * sample_rate), samples
    int nsrc, target;
    float ratio;                       // (float)10^(gain_nontarget/10); used only when rescale != 0
    int rescale;
    long long start, out_len;
};

// mean power of every source: power[j] = (float)(sum_i src_j[i]^2 / n_j), squares formed in fp32 as the reference's `sig ** 2` does
__global__ __launch_bounds__(256) void mix_power_kernel(MixArgs a, float *__restrict__ power) {
    __shared__ double part[256];
    const int j = blockIdx.x;
    const float *s = a.src + a.off[j];
    const long long n = a.off[j + 1] - a.off[j];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float v = s[i];
        acc += (double)(v * v);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) power[j] = n > 0 ? (float)(part[0] / (double)n) : 0.f;
}

__global__ __launch_bounds__(256) void mix_sum_kernel(MixArgs a, const float *__restrict__ power, float *__restrict__ out) {
    float gain[MIX_MAX_SRC];
#pragma unroll
    for (int j = 0; j < MIX_MAX_SRC; ++j) {
        gain[j] = 1.f;
        if (a.rescale && j < a.nsrc && j != a.target) {
            // the reference's 0-dim tensor expression, operation by operation in fp32: ((ratio * p_target) / p_j).sqrt()
            const float num = a.ratio * power[a.target];
            gain[j] = __fsqrt_rn(__fdiv_rn(num, power[j]));
        }
    }
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= a.out_len) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long p = a.start + i0 + e;        // position in the un-cropped mixture
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < MIX_MAX_SRC; ++j) {
            if (j >= a.nsrc) break;
            const long long q = p - a.delay[j], n = a.off[j + 1] - a.off[j];
            float x = 0.f;
            if (q >= 0 && q < n) {
                x = a.src[a.off[j] + q];
                if (a.rescale && j != a.target) x = x * gain[j];
            }
            acc = j == 0 ? x : acc + x;
        }
        v[e] = acc;
    }
    if (i0 + 4 <= a.out_len && (reinterpret_cast<uintptr_t>(out + i0) & 15) == 0) {
        *reinterpret_cast<float4 *>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int e = 0; e < 4 && i0 + e < a.out_len; ++e) out[i0 + e] = v[e];
    }
}

extern "C" {

size_t tsasr_mix_sources_workspace_bytes(void) { return MIX_MAX_SRC * sizeof(float); }

long long tsasr_mix_sources_out_len(const long long *host_off, const int *host_delay, int nsrc, long long start, long long duration) {
    long long n = 0;
    for (int j = 0; j < nsrc; ++j) n = std::max(n, host_off[j + 1] - host_off[j] + (long long)host_delay[j]);
    if (start >= n) return 0;
    return std::min(duration, n - start);
}

int tsasr_mix_sources(const float *src, const long long *host_off, const int *host_delay, int nsrc, int target, float ratio, int rescale,
                      long long start, long long duration, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(src && host_off && host_delay && out, "tsasr_mix_sources: null pointer");
    TSASR_CHECK_ARG(nsrc >= 1 && nsrc <= MIX_MAX_SRC, "tsasr_mix_sources: 1..%d sources (got %d)", MIX_MAX_SRC, nsrc);
    TSASR_CHECK_ARG(target >= 0 && target < nsrc, "tsasr_mix_sources: target index %d outside the %d sources", target, nsrc);
    TSASR_CHECK_ARG(start >= 0 && duration >= 0, "tsasr_mix_sources: negative window");
    TSASR_CHECK_ARG(!rescale || (workspace && workspace_bytes >= tsasr_mix_sources_workspace_bytes()), "tsasr_mix_sources: workspace too small");
    MixArgs a;
    a.src = src;
    for (int j = 0; j <= nsrc; ++j) {
        a.off[j] = host_off[j];
        TSASR_CHECK_ARG(j == 0 || host_off[j] >= host_off[j - 1], "tsasr_mix_sources: offsets must ascend");
    }
    for (int j = 0; j < nsrc; ++j) {
        TSASR_CHECK_ARG(host_delay[j] >= 0, "tsasr_mix_sources: negative delay");
        TSASR_CHECK_ARG(!rescale || host_off[j + 1] > host_off[j], "tsasr_mix_sources: empty source %d has no power to rescale by", j);
        a.delay[j] = host_delay[j];
    }
    a.nsrc = nsrc; a.target = target; a.ratio = ratio; a.rescale = rescale ? 1 : 0; a.start = start;
    a.out_len = tsasr_mix_sources_out_len(host_off, host_delay, nsrc, start, duration);
    if (a.out_len == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float *power = reinterpret_cast<float *>(workspace);
    if (a.rescale) {
        mix_power_kernel<<<nsrc, 256, 0, st>>>(a, power);
        TSASR_CHECK_LAUNCH("mix_power_kernel");
    }
    const long long groups = (a.out_len + 3) / 4;
    mix_sum_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, st>>>(a, power, out);
    TSASR_CHECK_LAUNCH("mix_sum_kernel");
    return 0;
}

}  // extern "C"
