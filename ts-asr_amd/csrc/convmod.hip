// Conformer convolution module core for gfx950: bias + GLU + depthwise Conv1d(K) + LayerNorm + LeakyReLU, fwd and bwd.
//
// Replaces the middle of ConvolutionModule.forward (vendor/speechbrain/speechbrain/lobes/models/transformer/Conformer.py:
// 101-115): `bottleneck` bias + nn.GLU(dim=1) (:76-82), depthwise nn.Conv1d(D, D, K, groups=D) with 'same' padding or
// causal pad+chomp (:68-71,84-93,108-110), `after_conv` LayerNorm + activation (:95-97). The two pointwise GEMMs around
// it are csrc/gemm.hip. In the reference this is two layout transposes, a GLU kernel, MIOpen's grouped conv
// (im2col + one tiny GEMM per channel group on ROCm: 384 launches per step), LayerNorm and LeakyReLU.
//
// Layout: channels-last [B, T, 2D] in, [B, T, D] out - time is the row index. Two launches per direction:
//   forward : (1) glu_dwconv_fwd   y2 -> c = dwconv(GLU(y2 + b2)) + conv_b   (io dtype; saved for the backward)
//             (2) tsasr_layernorm_fwd(c) with the fused LeakyReLU -> z, mean, rstd            (csrc/elementwise.hip)
//   backward: (1) tsasr_layernorm_bwd(dz; c) -> dc, dgamma, dbeta
//             (2) glu_dwconv_bwd   dc, y2 -> dy2 and per-workgroup partial rows of dconv_b, db2, dconv_w (reduced by colsum)
// The depthwise stage is independent per channel, so a workgroup takes 64 frames x 64 channels (+K-1 halo frames): the GLU
// tile (and in the backward the dc tile) sit in LDS as fp32 (24 / 48 KB -> 3-6 workgroups per CU), thread = (channel, group
// of 16 frames), filter taps in registers. An earlier version ran all of it in ONE kernel that owned every channel of 32
// frames (LayerNorm needs whole rows): 135 KB of LDS = one 4-wave workgroup per CU, every phase latency-bound (s_memtime
// stamps: 17 us tile load, 38 us row normalisation, 26 us taps) - 105 us per call against ~3 us of HBM traffic. Splitting
// the row-wise and the channel-wise halves costs one extra [B,T,D] round trip and runs each half at full occupancy.
#include <algorithm>

#include <stdlib.h>

#include "common.h"

#define CV_TT 64       // output frames per workgroup
#define CV_CH 64       // channels per workgroup
#define CV_FPT 16      // frames per thread (4 frame groups x 64 channels = 256 threads)

__device__ __forceinline__ float sigmoidf_fast(float x) { return 1.f / (1.f + __expf(-x)); }

// rows [row0, row0+R) x channels [c0, c0+64) of GLU(y2 + b2) of utterance b -> g_lds[R][64] (zero outside [0,T) x [0,D)).
// All of a thread's 16-byte pieces are requested before the first sigmoid (one memory round trip).
template <typename T, int R>
__device__ __forceinline__ void load_glu_tile(const T *__restrict__ y2, const float *__restrict__ b2, float *g_lds, int b, int Tn,
                                              int D, int c0, int row0) {
    constexpr int ITEMS = R * (CV_CH / 8), NIT = (ITEMS + 255) / 256;
    float a[NIT][8], g[NIT][8];
    bool live[NIT];
    // this thread's channel chunk is the same in every pass (256 % 8 == 0): its 8 + 8 bias values are fetched ONCE, with the tile
    // (read as `b2 ? b2[c + j] : 0` inside the GLU loop they were 48 guarded loads = 48 serialized round trips per workgroup)
    float ba[8], bg[8];
    {
        const int cb = min(c0 + (int)(threadIdx.x & 7) * 8, D - 8);
        if (b2) { ld8(b2 + cb, ba); ld8(b2 + D + cb, bg); }
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ba[j] = bg[j] = 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = threadIdx.x + u * 256, rr = i >> 3, c = c0 + (i & 7) * 8, t = row0 + rr;
        live[u] = i < ITEMS && t >= 0 && t < Tn && c < D;
        // always issued (frame / channel clamped): a guarded load would be waited for where it stands
        const T *p = y2 + ((size_t)b * Tn + min(max(t, 0), Tn - 1)) * 2 * D;
        ld8(p + min(c, D - 8), a[u]);
        ld8(p + D + min(c, D - 8), g[u]);
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = threadIdx.x + u * 256, rr = i >> 3, cl = (i & 7) * 8;
        if (i >= ITEMS) break;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = live[u] ? (a[u][j] + ba[j]) * sigmoidf_fast(g[u][j] + bg[j]) : 0.f;
        *reinterpret_cast<float4 *>(g_lds + rr * CV_CH + cl) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(g_lds + rr * CV_CH + cl + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const T *__restrict__ y2, const float *__restrict__ b2,
                                                             const float *__restrict__ cw, const float *__restrict__ cb,
                                                             T *__restrict__ c_out, int Tn, int D, int pad_l) {
    constexpr int R = CV_TT + K - 1;
    __shared__ __attribute__((aligned(16))) float g_lds[R * CV_CH];
    const int b = blockIdx.z, c0 = blockIdx.y * CV_CH, t0 = blockIdx.x * CV_TT;
    load_glu_tile<T, R>(y2, b2, g_lds, b, Tn, D, c0, t0 - pad_l);
    __syncthreads();
    const int ch = threadIdx.x & 63, fg = threadIdx.x >> 6, d = c0 + ch;
    if (d >= D) return;
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = cw[d * K + k];
    const float bias = cb[d];
    // this thread's 16 output frames read 16 + K - 1 consecutive GLU rows of ONE channel: fetched from LDS once into registers
    // (46 reads) instead of once per tap (496 reads: the tap loops were LDS-issue bound)
    float win[CV_FPT + K - 1];
#pragma unroll
    for (int j = 0; j < CV_FPT + K - 1; ++j) win[j] = g_lds[(fg * CV_FPT + j) * CV_CH + ch];
    float outv[CV_FPT];
#pragma unroll
    for (int s = 0; s < CV_FPT; ++s) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w[k] * win[s + k];
        outv[s] = acc;
    }
    // a whole tile inside the utterance (all but the last time tile) stores without guards: behind a per-frame guard every store was
    // waited for before the next (one data register, s_waitcnt vmcnt(0) in front of each of the 16) - 16 serialized round trips
    T *cp = c_out + ((size_t)b * Tn + t0 + fg * CV_FPT) * D + d;
    if (t0 + CV_TT <= Tn) {
#pragma unroll
        for (int s = 0; s < CV_FPT; ++s) st1(cp + (size_t)s * D, outv[s]);
    } else {
#pragma unroll
        for (int s = 0; s < CV_FPT; ++s)
            if (t0 + fg * CV_FPT + s < Tn) st1(cp + (size_t)s * D, outv[s]);
    }
}

// slab row of one workgroup-part (b, time tile), floats: [dconv_b D][db2 2D][dconv_w D*K]; every channel group fills its own columns
template <typename T, int K>
__global__ __launch_bounds__(256) void glu_dwconv_bwd_kernel(const T *__restrict__ dc, const T *__restrict__ y2,
                                                             const float *__restrict__ b2, const float *__restrict__ cw,
                                                             T *__restrict__ dy2, float *__restrict__ slab, int Tn, int D, int pad_l) {
    constexpr int R = CV_TT + K - 1, NP = K + 3;   // NP partial sums per channel: dconv_w[K], dconv_b, db2 (value half), db2 (gate half)
    constexpr int TILE = R * CV_CH, RED = 4 * CV_CH * NP;
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE > RED ? 2 * TILE : RED];
    float *g_lds = smem;           // GLU rows  [t0 - pad_l, +R)
    float *dc_lds = smem + TILE;   // dc rows   [t0 - (K-1-pad_l), +R)
    const int b = blockIdx.z, c0 = blockIdx.y * CV_CH, t0 = blockIdx.x * CV_TT;
    const int dc_row0 = t0 - (K - 1 - pad_l);
    {   // dc tile (zero outside the utterance / channel range): 16-byte pieces, requested together with the GLU operands
        constexpr int ITEMS = R * (CV_CH / 8), NIT = (ITEMS + 255) / 256;
        float v[NIT][8];
        bool live[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = threadIdx.x + u * 256, rr = i >> 3, c = c0 + (i & 7) * 8, t = dc_row0 + rr;
            live[u] = i < ITEMS && t >= 0 && t < Tn && c < D;
            ld8(dc + ((size_t)b * Tn + min(max(t, 0), Tn - 1)) * D + min(c, D - 8), v[u]);
        }
        load_glu_tile<T, R>(y2, b2, g_lds, b, Tn, D, c0, t0 - pad_l);
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = threadIdx.x + u * 256, rr = i >> 3, cl = (i & 7) * 8;
            if (i >= ITEMS) break;
            *reinterpret_cast<float4 *>(dc_lds + rr * CV_CH + cl) = live[u] ? make_float4(v[u][0], v[u][1], v[u][2], v[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dc_lds + rr * CV_CH + cl + 4) = live[u] ? make_float4(v[u][4], v[u][5], v[u][6], v[u][7]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const int ch = threadIdx.x & 63, fg = threadIdx.x >> 6, d = c0 + ch;
    const bool ch_ok = d < D;
    float dw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dw[k] = 0.f;
    float dcb = 0.f, dba = 0.f, dbb = 0.f;
    if (ch_ok) {
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = cw[d * K + k];
        const float ba = b2 ? b2[d] : 0.f, bb = b2 ? b2[D + d] : 0.f;
        const int own_off = K - 1 - pad_l;   // dc_lds row of frame t0
        float a_in[CV_FPT], g_in[CV_FPT];    // GEMM outputs (value / gate halves) of this thread's frames, requested before the tap loops
#pragma unroll
        for (int s = 0; s < CV_FPT; ++s) {
            const T *p = y2 + ((size_t)b * Tn + min(t0 + fg * CV_FPT + s, Tn - 1)) * 2 * D;
            a_in[s] = ld1(p + d);
            g_in[s] = ld1(p + D + d);
        }
        // register windows (see the forward kernel): GLU rows and dc rows [fg*16, fg*16 + 16 + K - 1) of this channel
        float gw[CV_FPT + K - 1], dcw[CV_FPT + K - 1];
#pragma unroll
        for (int j = 0; j < CV_FPT + K - 1; ++j) {
            gw[j] = g_lds[(fg * CV_FPT + j) * CV_CH + ch];
            dcw[j] = dc_lds[(fg * CV_FPT + j) * CV_CH + ch];
        }
        float dav[CV_FPT], dbv[CV_FPT];
#pragma unroll
        for (int s = 0; s < CV_FPT; ++s) {
            const int tl = fg * CV_FPT + s, t = t0 + tl;
            // dW[k] += dc[t] * g[t + k - pad_l] ; dconv_b += dc[t] ; dg[t] = sum_k w[k] * dc[t - k + pad_l]
            const float dct = t < Tn ? dc_lds[(own_off + tl) * CV_CH + ch] : 0.f;   // (own_off is a run-time row: one LDS read)
            dcb += dct;
            float dg = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                dw[k] += dct * gw[s + k];
                dg += w[k] * dcw[s + (K - 1) - k];
            }
            const float a = a_in[s] + ba, sg = sigmoidf_fast(g_in[s] + bb);
            const float da = t < Tn ? dg * sg : 0.f, db = t < Tn ? dg * a * sg * (1.f - sg) : 0.f;
            dav[s] = da;
            dbv[s] = db;
            dba += da;
            dbb += db;
        }
        // stores without per-frame guards when the whole tile lies inside the utterance (see the forward kernel)
        T *q = dy2 + ((size_t)b * Tn + t0 + fg * CV_FPT) * 2 * D + d;
        if (t0 + CV_TT <= Tn) {
#pragma unroll
            for (int s = 0; s < CV_FPT; ++s) {
                st1(q + (size_t)s * 2 * D, dav[s]);
                st1(q + (size_t)s * 2 * D + D, dbv[s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < CV_FPT; ++s)
                if (t0 + fg * CV_FPT + s < Tn) {
                    st1(q + (size_t)s * 2 * D, dav[s]);
                    st1(q + (size_t)s * 2 * D + D, dbv[s]);
                }
        }
    }
    __syncthreads();   // tiles are dead: reuse them for the cross-frame-group reduction [fg][NP][64]
    float *red = smem;
#pragma unroll
    for (int k = 0; k < K; ++k) red[(fg * NP + k) * CV_CH + ch] = dw[k];
    red[(fg * NP + K) * CV_CH + ch] = dcb;
    red[(fg * NP + K + 1) * CV_CH + ch] = dba;
    red[(fg * NP + K + 2) * CV_CH + ch] = dbb;
    __syncthreads();
    float *my = slab + (size_t)(blockIdx.z * gridDim.x + blockIdx.x) * (size_t)(D * (K + 3));
    for (int i = threadIdx.x; i < NP * CV_CH; i += 256) {
        const int q = i >> 6, c = i & 63;
        if (c0 + c >= D) continue;
        const float v = red[(0 * NP + q) * CV_CH + c] + red[(1 * NP + q) * CV_CH + c] + red[(2 * NP + q) * CV_CH + c] + red[(3 * NP + q) * CV_CH + c];
        if (q < K) my[3 * D + (size_t)(c0 + c) * K + q] = v;
        else if (q == K) my[c0 + c] = v;
        else if (q == K + 1) my[D + c0 + c] = v;
        else my[2 * D + c0 + c] = v;
    }
}

// =====================================================================================================================
// One launch per direction for bf16 rows of D = 256 channels (the recipes' d_model): a workgroup owns ALL channels of 32 frames
// (+ K - 1 halo frames), so the row-wise half (LayerNorm + LeakyReLU) runs on the tile the channel-wise half left in LDS.
//   The pair above is two launches forward (7.4 + 5.4 us at B.T' = 8000) and three backward (10 + 20 us + the LayerNorm's column
//   reduction) in a chain whose length is kernel count x per-kernel latency; the first one-kernel version
//   (header) failed on 135 KB of LDS behind FOUR waves. Here: 1024 threads forward (thread = channel x 8 frames, 77 live
//   registers), 512 backward (channel x 16 frames: the two tap loops need ~220), the GLU tile's memory reused for the c tile, the
//   backward recomputing the LayerNorm backward of its halo rows (62 rows for 32 owned) instead of exchanging dc through HBM.
//   Bit-compatible with the pair where the pair rounds: c and dc pass through bf16 exactly where the pair stored them, the row
//   statistics use layernorm_fwd_kernel<bf16, 32, 1>'s lane layout and sums - z, c_save, mean, rstd and dy2 are bit-identical;
//   the parameter gradients are summed per workgroup in a different order (fp32).
// =====================================================================================================================
#define CF_TT 32
#define CF_D 256

template <int R, int NT>
__device__ __forceinline__ void cf_load_glu_tile(const bf16_t *__restrict__ y2, const float *__restrict__ b2, float *g_lds, int b, int Tn, int row0) {
    constexpr int ITEMS = R * (CF_D / 8), NIT = (ITEMS + NT - 1) / NT;
    float a[NIT][8], g[NIT][8], ba[8], bg[8];
    bool live[NIT];
    const int cl = (int)(threadIdx.x & 31) * 8;     // NT % 32 == 0: the same channel chunk in every pass
    if (b2) { ld8(b2 + cl, ba); ld8(b2 + CF_D + cl, bg); }
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) ba[j] = bg[j] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = threadIdx.x + u * NT, t = row0 + (i >> 5);
        live[u] = i < ITEMS && t >= 0 && t < Tn;
        const bf16_t *p = y2 + ((size_t)b * Tn + min(max(t, 0), Tn - 1)) * 2 * CF_D;
        ld8(p + cl, a[u]);
        ld8(p + CF_D + cl, g[u]);
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = threadIdx.x + u * NT, rr = i >> 5;
        if (i >= ITEMS) break;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = live[u] ? (a[u][j] + ba[j]) * sigmoidf_fast(g[u][j] + bg[j]) : 0.f;
        *reinterpret_cast<float4 *>(g_lds + rr * CF_D + cl) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(g_lds + rr * CF_D + cl + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

template <int K, int NT>
__device__ __forceinline__ void cf_stage_filters(const float *__restrict__ cw, float *wt) {
    constexpr int PIECES = CF_D * K / 4, NIT = (PIECES + NT - 1) / NT;
    float4 v[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) v[u] = *reinterpret_cast<const float4 *>(cw + 4 * min((int)threadIdx.x + u * NT, PIECES - 1));
#pragma unroll
    for (int u = 0; u < NIT; ++u)
        if ((int)threadIdx.x + u * NT < PIECES) *reinterpret_cast<float4 *>(wt + 4 * (threadIdx.x + u * NT)) = v[u];
}

template <int K>
__global__ __launch_bounds__(1024) void convmod_fwd_fused_kernel(const bf16_t *__restrict__ y2, const float *__restrict__ b2, const float *__restrict__ cw,
                                                                 const float *__restrict__ cb, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 bf16_t *__restrict__ z, bf16_t *__restrict__ c_save, float *__restrict__ mean,
                                                                 float *__restrict__ rstd, int Tn, int pad_l, float eps, float slope) {
    constexpr int R = CF_TT + K - 1, FPT = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *g_lds = smem;                 // [R][256]: first the GLU rows, then the 32 rows of c (K = 31: 62 KB)
    float *wt = smem + R * CF_D;         // the filters [256][K] as they lie in memory (31 KB)
    const int b = blockIdx.y, t0 = blockIdx.x * CF_TT, tid = threadIdx.x, ch = tid & 255, fg = tid >> 8;
    // filters: coalesced 16-byte pieces through LDS (read as cw[ch * K + k] every wave-load touched 64 lines for 256 bytes: 16 waves x 31 such
    // loads were ~30k line requests per workgroup, several microseconds of the texture path)
    cf_stage_filters<K, 1024>(cw, wt);
    const float bias = cb[ch];
    cf_load_glu_tile<R, 1024>(y2, b2, g_lds, b, Tn, t0 - pad_l);
    __syncthreads();
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = wt[ch * K + k];     // lane stride K floats: odd K, no bank conflicts
    float win[FPT + K - 1];
#pragma unroll
    for (int j = 0; j < FPT + K - 1; ++j) win[j] = g_lds[(fg * FPT + j) * CF_D + ch];
    float outv[FPT];
#pragma unroll
    for (int s = 0; s < FPT; ++s) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w[k] * win[s + k];
        outv[s] = acc;
    }
    __syncthreads();     // every window is in registers: the tile's first 32 rows become c, rounded as the pair stores it
#pragma unroll
    for (int s = 0; s < FPT; ++s) g_lds[(fg * FPT + s) * CF_D + ch] = (float)(bf16_t)outv[s];
    __syncthreads();
    // rows: 16 waves x 2 half-waves = the 32 frames, 8 channels per lane (layernorm_fwd_kernel<bf16, 32, 1>'s layout and order of sums)
    const int lane = tid & 63, l = lane & 31, rr = (tid >> 6) * 2 + (lane >> 5), t = t0 + rr;
    float v[8];
    {
        const float4 v0 = *reinterpret_cast<const float4 *>(g_lds + rr * CF_D + 8 * l), v1 = *reinterpret_cast<const float4 *>(g_lds + rr * CF_D + 8 * l + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    float g[8], be[8];
    ld8(gamma + 8 * l, g);
    ld8(beta + 8 * l, be);
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sm += v[j];
    const float mu = half_wave_sum(sm) / CF_D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
    const float rs = rsqrtf(half_wave_sum(q) / CF_D + eps);
    if (t >= Tn) return;
    const size_t row = (size_t)b * Tn + t;
    if (l == 0) { mean[row] = mu; rstd[row] = rs; }
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float y = (v[j] - mu) * rs * g[j] + be[j];
        if (slope >= 0.f) y = lrelu(y, slope);
        o[j] = y;
    }
    st8(z + row * CF_D + 8 * l, o);
    st8(c_save + row * CF_D + 8 * l, v);
}

// slab row of one workgroup (b, 32-frame tile), floats, in the order of tsasr_convmod_bwd's dparams: [dgamma D][dbeta D][dconv_b D][db2 2D][dconv_w D*K]
template <int K>
__global__ __launch_bounds__(512) void convmod_bwd_fused_kernel(const bf16_t *__restrict__ dz, const bf16_t *__restrict__ y2, const float *__restrict__ b2,
                                                                const float *__restrict__ cw, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                const bf16_t *__restrict__ c_save, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                bf16_t *__restrict__ dy2, float *__restrict__ slab, int Tn, int pad_l, float slope) {
    constexpr int R = CF_TT + K - 1, NP = K + 3, TILE = R * CF_D, FPT = 16, NPASS = (R + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *g_lds = smem;           // GLU rows  [t0 - pad_l, +R)
    float *dc_lds = smem + TILE;   // dc rows   [t0 - (K-1-pad_l), +R): the LayerNorm backward of every row this tile's taps touch
    float *wt = smem + 2 * TILE;   // the filters [256][K] (see the forward kernel)
    const int b = blockIdx.y, t0 = blockIdx.x * CF_TT, tid = threadIdx.x, lane = tid & 63, l = lane & 31, hw = (tid >> 6) * 2 + (lane >> 5);
    const int own_off = K - 1 - pad_l, dc_row0 = t0 - own_off;
    const int ch = tid & 255, fg = tid >> 8;
    // ---- every operand of both halves requested up front
    float gam[8], bet[8];
    ld8(gamma + 8 * l, gam);
    ld8(beta + 8 * l, bet);
    float xv[NPASS][8], dv[NPASS][8], mu[NPASS], rsd[NPASS];
    bool live[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int rr = hw + 16 * ps, t = dc_row0 + rr;
        live[ps] = rr < R && t >= 0 && t < Tn;
        const size_t row = (size_t)b * Tn + min(max(t, 0), Tn - 1);
        ld8(c_save + row * CF_D + 8 * l, xv[ps]);
        ld8(dz + row * CF_D + 8 * l, dv[ps]);
        mu[ps] = mean[row];
        rsd[ps] = rstd[row];
    }
    cf_stage_filters<K, 512>(cw, wt);
    const float ba = b2 ? b2[ch] : 0.f, bb = b2 ? b2[CF_D + ch] : 0.f;
    cf_load_glu_tile<R, 512>(y2, b2, g_lds, b, Tn, t0 - pad_l);
    // ---- row-wise half: dc = LayerNorm'(LeakyReLU'(dz)) for the tile's R rows; dgamma / dbeta from the 32 owned rows only
    float ag[8], abt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[j] = abt[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int rr = hw + 16 * ps;
        const bool own = live[ps] && rr >= own_off && rr < own_off + CF_TT;
        float xh[8], gdy[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float h = (xv[ps][j] - mu[ps]) * rsd[ps];
            float d = dv[ps][j];
            if (slope >= 0.f && (h * gam[j] + bet[j]) <= 0.f) d *= slope;
            xh[j] = h;
            ag[j] += own ? d * h : 0.f;
            abt[j] += own ? d : 0.f;
            const float gd = d * gam[j];
            gdy[j] = gd;
            s1 += gd;
            s2 += gd * h;
        }
        const float m1 = half_wave_sum(s1) / CF_D, m2 = half_wave_sum(s2) / CF_D;
        if (rr < R) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = live[ps] ? (float)(bf16_t)(rsd[ps] * (gdy[j] - m1 - xh[j] * m2)) : 0.f;
            *reinterpret_cast<float4 *>(dc_lds + rr * CF_D + 8 * l) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4 *>(dc_lds + rr * CF_D + 8 * l + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
    float a_in[FPT], g_in[FPT];      // this thread's own GEMM outputs (L2 hits: the tile load fetched the lines), in flight across the barrier
#pragma unroll
    for (int s = 0; s < FPT; ++s) {
        const bf16_t *p = y2 + ((size_t)b * Tn + min(t0 + fg * FPT + s, Tn - 1)) * 2 * CF_D;
        a_in[s] = ld1(p + ch);
        g_in[s] = ld1(p + CF_D + ch);
    }
    __syncthreads();
    // ---- channel-wise half (glu_dwconv_bwd_kernel's, on 32 frames x 256 channels)
    float dw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dw[k] = 0.f;
    float dcb = 0.f, dba = 0.f, dbb = 0.f;
    {   // data gradient first (filter + dc window live), then the filter gradient (GLU window + 31 sums live): one loop held ~220 registers
        float dav[FPT], dbv[FPT];
        {
            float w[K];
#pragma unroll
            for (int k = 0; k < K; ++k) w[k] = wt[ch * K + k];
            float dcw[FPT + K - 1];
#pragma unroll
            for (int j = 0; j < FPT + K - 1; ++j) dcw[j] = dc_lds[(fg * FPT + j) * CF_D + ch];
#pragma unroll
            for (int s = 0; s < FPT; ++s) {
                const int t = t0 + fg * FPT + s;
                float dg = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) dg += w[k] * dcw[s + (K - 1) - k];
                const float a = a_in[s] + ba, sg = sigmoidf_fast(g_in[s] + bb);
                const float da = t < Tn ? dg * sg : 0.f, db = t < Tn ? dg * a * sg * (1.f - sg) : 0.f;
                dav[s] = da;
                dbv[s] = db;
                dba += da;
                dbb += db;
            }
        }
        bf16_t *q = dy2 + ((size_t)b * Tn + t0 + fg * FPT) * 2 * CF_D + ch;
        if (t0 + CF_TT <= Tn) {
#pragma unroll
            for (int s = 0; s < FPT; ++s) {
                st1(q + (size_t)s * 2 * CF_D, dav[s]);
                st1(q + (size_t)s * 2 * CF_D + CF_D, dbv[s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < FPT; ++s)
                if (t0 + fg * FPT + s < Tn) {
                    st1(q + (size_t)s * 2 * CF_D, dav[s]);
                    st1(q + (size_t)s * 2 * CF_D + CF_D, dbv[s]);
                }
        }
        float gw[FPT + K - 1];
#pragma unroll
        for (int j = 0; j < FPT + K - 1; ++j) gw[j] = g_lds[(fg * FPT + j) * CF_D + ch];
#pragma unroll
        for (int s = 0; s < FPT; ++s) {
            const int tl = fg * FPT + s, t = t0 + tl;
            const float dct = t < Tn ? dc_lds[(own_off + tl) * CF_D + ch] : 0.f;
            dcb += dct;
#pragma unroll
            for (int k = 0; k < K; ++k) dw[k] += dct * gw[s + k];
        }
    }
    __syncthreads();   // tiles are dead: [2 frame groups][NP][256] channel partials, then [16 half-waves][2][256] LayerNorm partials
    float *red = smem, *lnred = smem + 2 * NP * CF_D;
    static_assert(2 * NP * CF_D + 16 * 2 * CF_D <= 2 * TILE, "reduction buffers fit in the tiles");
#pragma unroll
    for (int k = 0; k < K; ++k) red[(fg * CF_D + ch) * NP + k] = dw[k];     // [frame group][channel][NP]: the slab's dconv_w rows are then contiguous runs
    red[(fg * CF_D + ch) * NP + K] = dcb;
    red[(fg * CF_D + ch) * NP + K + 1] = dba;
    red[(fg * CF_D + ch) * NP + K + 2] = dbb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lnred[(hw * 2 + 0) * CF_D + 8 * l + j] = ag[j];
        lnred[(hw * 2 + 1) * CF_D + 8 * l + j] = abt[j];
    }
    __syncthreads();
    float *my = slab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(CF_D * (K + 5));
    {   // dgamma | dbeta: thread = (which, channel), 16 half-wave partials in a fixed order
        const int which = tid >> 8;
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < 16; ++h) a += lnred[(h * 2 + which) * CF_D + ch];
        my[which * CF_D + ch] = a;
    }
    for (int i = tid; i < NP * CF_D; i += 512) {
        const int c = i / NP, q = i - c * NP;
        const float v = red[i] + red[CF_D * NP + i];
        if (q < K) my[5 * CF_D + c * K + q] = v;
        else my[(q - K + 2) * CF_D + c] = v;      // dconv_b at 2D, db2 (value half) at 3D, db2 (gate half) at 4D
    }
}

static bool convmod_fused(int io_dtype, int D) {      // TSASR_CONVMOD_FUSED=0: the two / three-launch pair (A/B; read per call: the tests compare both in one process)
    const char *e = getenv("TSASR_CONVMOD_FUSED");
    return (!e || e[0] != '0') && io_dtype == TSASR_BF16 && D == CF_D;
}
static size_t fused_slab_bytes(int B, int T, int D, int K) { return align_up((size_t)B * cdiv(T, CF_TT) * D * (K + 5) * sizeof(float), 256); }

template <int K>
static void launch_fwd_fused(const void *y2, const float *b2, const float *cw, const float *cb, const float *gamma, const float *beta, void *z, void *cs,
                             float *mean, float *rstd, int B, int Tn, int pad_l, float eps, float slope, hipStream_t st) {
    constexpr size_t lds = ((size_t)(CF_TT + K - 1) * CF_D + (size_t)CF_D * K) * sizeof(float);
    static const bool once = [] { (void)hipFuncSetAttribute((const void *)convmod_fwd_fused_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); return true; }();
    (void)once;
    convmod_fwd_fused_kernel<K><<<dim3(cdiv(Tn, CF_TT), B), 1024, lds, st>>>((const bf16_t *)y2, b2, cw, cb, gamma, beta, (bf16_t *)z, (bf16_t *)cs, mean, rstd, Tn,
                                                                          pad_l, eps, slope);
}
template <int K>
static void launch_bwd_fused(const void *dz, const void *y2, const float *b2, const float *cw, const float *gamma, const float *beta, const void *cs,
                             const float *mean, const float *rstd, void *dy2, float *slab, int B, int Tn, int pad_l, float slope, hipStream_t st) {
    constexpr size_t lds = ((size_t)2 * (CF_TT + K - 1) * CF_D + (size_t)CF_D * K) * sizeof(float);
    static const bool once = [] { (void)hipFuncSetAttribute((const void *)convmod_bwd_fused_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); return true; }();
    (void)once;
    convmod_bwd_fused_kernel<K><<<dim3(cdiv(Tn, CF_TT), B), 512, lds, st>>>((const bf16_t *)dz, (const bf16_t *)y2, b2, cw, gamma, beta, (const bf16_t *)cs, mean, rstd,
                                                                           (bf16_t *)dy2, slab, Tn, pad_l, slope);
}

template <typename T, int K>
static void launch_fwd(const void *y2, const float *b2, const float *cw, const float *cb, void *cs, int B, int Tn, int D, int pad_l,
                       hipStream_t st) {
    glu_dwconv_fwd_kernel<T, K><<<dim3(cdiv(Tn, CV_TT), cdiv(D, CV_CH), B), 256, 0, st>>>((const T *)y2, b2, cw, cb, (T *)cs, Tn, D, pad_l);
}

template <typename T, int K>
static void launch_bwd(const void *dc, const void *y2, const float *b2, const float *cw, void *dy2, float *slab, int B, int Tn, int D,
                       int pad_l, hipStream_t st) {
    glu_dwconv_bwd_kernel<T, K><<<dim3(cdiv(Tn, CV_TT), cdiv(D, CV_CH), B), 256, 0, st>>>((const T *)dc, (const T *)y2, b2, cw, (T *)dy2, slab, Tn, D, pad_l);
}

static size_t slab_bytes(int B, int T, int D, int K) { return align_up((size_t)B * cdiv(T, CV_TT) * D * (K + 3) * sizeof(float), 256); }
static size_t dc_bytes(int B, int T, int D) { return align_up((size_t)B * T * D * sizeof(float), 256); }

extern "C" {

/* y2 [B,T,2D] (bottleneck GEMM output WITHOUT its bias; b2 = that bias [2D] or NULL), conv_w [D,K] (the [D,1,K] parameter),
 * conv_b [D], gamma/beta [D]  ->  z [B,T,D] = LeakyReLU(LN(dwconv(GLU(y2+b2)))) ; c_save [B,T,D], mean/rstd [B*T] kept for bwd. */
int tsasr_convmod_fwd(const void *y2, const float *b2, const float *conv_w, const float *conv_b, const float *gamma,
                      const float *beta, void *z, void *c_save, float *mean, float *rstd, int B, int T, int D, int K, int causal,
                      float eps, float slope, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(y2 && conv_w && conv_b && gamma && beta && z && c_save && mean && rstd, "tsasr_convmod_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= 2048, "tsasr_convmod_fwd: D=%d must be a multiple of 8 and <= 2048", D);
    TSASR_CHECK_ARG(K == 31 || K == 15 || K == 7 || K == 3, "tsasr_convmod_fwd: kernel size %d not instantiated (31, 15, 7, 3)", K);
    TSASR_CHECK_ARG(io_dtype == TSASR_F32 || io_dtype == TSASR_BF16, "tsasr_convmod_fwd: bad io_dtype %d", io_dtype);
    const int pad_l = causal ? K - 1 : (K - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    if (convmod_fused(io_dtype, D)) {
#define CM_FF(KK) launch_fwd_fused<KK>(y2, b2, conv_w, conv_b, gamma, beta, z, c_save, mean, rstd, B, T, pad_l, eps, slope, st)
        if (K == 31) CM_FF(31); else if (K == 15) CM_FF(15); else if (K == 7) CM_FF(7); else CM_FF(3);
        TSASR_CHECK_LAUNCH("tsasr_convmod_fwd");
        return 0;
    }
#define CM_F(TT, KK) launch_fwd<TT, KK>(y2, b2, conv_w, conv_b, c_save, B, T, D, pad_l, st)
#define CM_FK(TT) do { if (K == 31) CM_F(TT, 31); else if (K == 15) CM_F(TT, 15); else if (K == 7) CM_F(TT, 7); else CM_F(TT, 3); } while (0)
    if (io_dtype == TSASR_F32) CM_FK(float);
    else CM_FK(bf16_t);
    TSASR_CHECK_LAUNCH("tsasr_convmod_fwd");
    return tsasr_layernorm_fwd(c_save, gamma, beta, z, mean, rstd, (long long)B * T, D, eps, slope, io_dtype, stream);
}

size_t tsasr_convmod_bwd_workspace_bytes(int B, int T, int D, int K) {
    return std::max(slab_bytes(B, T, D, K) + dc_bytes(B, T, D) + tsasr_layernorm_bwd_workspace_bytes((long long)B * T, D), fused_slab_bytes(B, T, D, K));
}

/* grads: dy2 [B,T,2D]; dparams fp32 packed [dgamma D | dbeta D | dconv_b D | db2 2D | dconv_w D*K] (OVERWRITTEN). */
int tsasr_convmod_bwd(const void *dz, const void *y2, const float *b2, const float *conv_w, const float *gamma, const float *beta,
                      const void *c_save, const float *mean, const float *rstd, void *dy2, float *dparams, int B, int T, int D,
                      int K, int causal, float slope, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dz && y2 && conv_w && gamma && beta && c_save && mean && rstd && dy2 && dparams && workspace, "tsasr_convmod_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= 2048, "tsasr_convmod_bwd: bad D=%d", D);
    TSASR_CHECK_ARG(K == 31 || K == 15 || K == 7 || K == 3, "tsasr_convmod_bwd: kernel size %d not instantiated", K);
    TSASR_CHECK_ARG(io_dtype == TSASR_F32 || io_dtype == TSASR_BF16, "tsasr_convmod_bwd: bad io_dtype %d", io_dtype);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_convmod_bwd_workspace_bytes(B, T, D, K), "tsasr_convmod_bwd: workspace too small");
    const int pad_l = causal ? K - 1 : (K - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    float *slab = (float *)workspace;
    if (convmod_fused(io_dtype, D)) {
#define CM_BF(KK) launch_bwd_fused<KK>(dz, y2, b2, conv_w, gamma, beta, c_save, mean, rstd, dy2, slab, B, T, pad_l, slope, st)
        if (K == 31) CM_BF(31); else if (K == 15) CM_BF(15); else if (K == 7) CM_BF(7); else CM_BF(3);
        const int widthf = D * (K + 5);
        tsasr_reduce_submit(slab, dparams, widthf, B * cdiv(T, CF_TT), widthf, 0, st);
        TSASR_CHECK_LAUNCH("tsasr_convmod_bwd");
        return 0;
    }
    void *dc = (char *)workspace + slab_bytes(B, T, D, K);
    void *ln_ws = (char *)dc + dc_bytes(B, T, D);
    const size_t ln_ws_bytes = tsasr_layernorm_bwd_workspace_bytes((long long)B * T, D);
    // (1) LayerNorm + LeakyReLU backward, row-wise: dc, dgamma, dbeta
    const int rc = tsasr_layernorm_bwd(dz, c_save, gamma, beta, mean, rstd, dc, dparams, dparams + D, (long long)B * T, D, slope, io_dtype,
                                       ln_ws, ln_ws_bytes, stream);
    if (rc) return rc;
    // (2) depthwise conv + GLU backward, channel-wise
#define CM_B(TT, KK) launch_bwd<TT, KK>(dc, y2, b2, conv_w, dy2, slab, B, T, D, pad_l, st)
#define CM_BK(TT) do { if (K == 31) CM_B(TT, 31); else if (K == 15) CM_B(TT, 15); else if (K == 7) CM_B(TT, 7); else CM_B(TT, 3); } while (0)
    if (io_dtype == TSASR_F32) CM_BK(float);
    else CM_BK(bf16_t);
    const int width = D * (K + 3), nparts = B * cdiv(T, CV_TT);
    tsasr_reduce_submit(slab, dparams + 2 * D, width, nparts, width, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_convmod_bwd");
    return 0;
}

}  // extern "C"
