// Conformer convolution module core for gfx950: bias + GLU + depthwise Conv1d(K) + LayerNorm + LeakyReLU, fwd and bwd.
//
// Replaces the middle of ConvolutionModule.forward (vendor/speechbrain/speechbrain/lobes/models/transformer/Conformer.py:
// 101-115): `bottleneck` bias + nn.GLU(dim=1) (:76-82), depthwise nn.Conv1d(D, D, K, groups=D) with 'same' padding or
// causal pad+chomp (:68-71,84-93,108-110), `after_conv` LayerNorm + activation (:95-97). The two pointwise GEMMs around
// it stay library GEMMs. In the reference this is two layout transposes, a GLU kernel, MIOpen's grouped conv
// (im2col + one tiny GEMM per channel group on ROCm: 384 launches per step), LayerNorm and LeakyReLU.
//
// Layout: channels-last [B, T, 2D] in, [B, T, D] out - time is the row index, so a workgroup owning TT consecutive
// frames of one utterance reads (TT + K - 1) full rows (coalesced 16-byte accesses along the channel axis; the K-1
// halo rows are the only re-read), keeps the GLU tile in LDS (fp32, row stride D: lane = channel -> conflict-free),
// runs the K taps out of LDS with the filter in registers, and normalises each frame with one wave (LayerNorm needs all
// channels of a frame, which is why channels are not split across workgroups).
// Backward recomputes GLU from the saved GEMM output, reads the saved pre-norm conv output c (io dtype) and the row
// statistics, and produces dy2 plus per-workgroup partial rows of every parameter gradient (reduced by colsum).
#include "common.h"

#define CM_TT 32       // output frames per workgroup
#define CM_MAXD 256    // channels handled by one workgroup (thread = channel in the conv phases)

__device__ __forceinline__ float sigmoidf_fast(float x) { return 1.f / (1.f + __expf(-x)); }

// rows [row0, row0+nrows) of the GLU output of utterance b into LDS (zero outside [0,T))
template <typename T>
__device__ __forceinline__ void load_glu_tile(const T *__restrict__ y2, const float *__restrict__ b2, float *g_lds, int b,
                                              int Tn, int D, int row0, int nrows) {
    const int vec_per_row = D / 8;
    for (int i = threadIdx.x; i < nrows * vec_per_row; i += 256) {
        const int rr = i / vec_per_row, c = (i % vec_per_row) * 8;
        const int t = row0 + rr;
        float o[8];
        if (t >= 0 && t < Tn) {
            float a[8], g[8];
            const T *p = y2 + ((size_t)b * Tn + t) * 2 * D;
            ld8(p + c, a);
            ld8(p + D + c, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (a[j] + (b2 ? b2[c + j] : 0.f)) * sigmoidf_fast(g[j] + (b2 ? b2[D + c + j] : 0.f));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) g_lds[rr * D + c + j] = o[j];
    }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void convmod_fwd_kernel(const T *__restrict__ y2, const float *__restrict__ b2,
                                                          const float *__restrict__ cw, const float *__restrict__ cb,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          T *__restrict__ z, T *__restrict__ c_save, float *__restrict__ mean,
                                                          float *__restrict__ rstd, int Tn, int D, int pad_l, float eps,
                                                          float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *g_lds = smem;                          // [(TT+K-1)][D]
    float *c_lds = smem + (CM_TT + K - 1) * D;    // [TT][D]
    const int b = blockIdx.y, t0 = blockIdx.x * CM_TT;
    load_glu_tile<T>(y2, b2, g_lds, b, Tn, D, t0 - pad_l, CM_TT + K - 1);
    __syncthreads();
    const int d = threadIdx.x;
    if (d < D) {
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = cw[d * K + k];
        const float bias = cb[d];
        for (int t = 0; t < CM_TT; ++t) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < K; ++k) acc += w[k] * g_lds[(t + k) * D + d];
            c_lds[t * D + d] = acc;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wave; t < CM_TT; t += 4) {
        const int tt = t0 + t;
        if (tt >= Tn) break;
        float v[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < D ? c_lds[t * D + c] : 0.f;
            s += v[j];
        }
        const float mu = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < D) { const float dd = v[j] - mu; q += dd * dd; }
        }
        const float rs = rsqrtf(wave_sum(q) / D + eps);
        const size_t row = (size_t)b * Tn + tt;
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                st1(c_save + row * D + c, v[j]);
                st1(z + row * D + c, lrelu((v[j] - mu) * rs * gamma[c] + beta[c], slope));
            }
        }
    }
}

// slab layout per workgroup (floats): [dgamma D][dbeta D][dcb D][db2 2D][dcw D*K]
template <typename T, int K>
__global__ __launch_bounds__(256) void convmod_bwd_kernel(const T *__restrict__ dz, const T *__restrict__ y2,
                                                          const float *__restrict__ b2, const float *__restrict__ cw,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const T *__restrict__ c_save, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, T *__restrict__ dy2,
                                                          float *__restrict__ slab, int Tn, int D, int pad_l, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = CM_TT + K - 1;
    float *g_lds = smem;               // GLU rows  [t0 - pad_l, t0 - pad_l + R)
    float *dc_lds = smem + R * D;      // dc rows   [t0 - (K-1-pad_l), ... + R)
    float *red = smem + 2 * R * D;     // [4][2][D] cross-wave combine of dgamma/dbeta
    const int b = blockIdx.y, t0 = blockIdx.x * CM_TT;
    const int g_row0 = t0 - pad_l, dc_row0 = t0 - (K - 1 - pad_l);
    load_glu_tile<T>(y2, b2, g_lds, b, Tn, D, g_row0, R);

    // ---- LayerNorm(+LeakyReLU) backward per frame -> dc rows (halo included); dgamma/dbeta over OWN frames only
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, abt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int rr = wave; rr < R; rr += 4) {
        const int t = dc_row0 + rr;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < Tn) {  // wave-uniform
            const size_t row = (size_t)b * Tn + t;
            const float mu = mean[row], rs = rstd[row];
            const bool own = (t >= t0) && (t < t0 + CM_TT);
            float xh[4], gd[4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane + 64 * j;
                xh[j] = gd[j] = 0.f;
                if (c < D) {
                    const float h = (ld1(c_save + row * D + c) - mu) * rs;
                    float dd = ld1(dz + row * D + c);
                    const float gm = gamma[c];
                    if (h * gm + beta[c] <= 0.f) dd *= slope;
                    if (own) { ag[j] += dd * h; abt[j] += dd; }
                    xh[j] = h;
                    gd[j] = dd * gm;
                    s1 += gd[j];
                    s2 += gd[j] * h;
                }
            }
            const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (gd[j] - m1 - xh[j] * m2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < D) dc_lds[rr * D + c] = o[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < D) { red[(wave * 2 + 0) * D + c] = ag[j]; red[(wave * 2 + 1) * D + c] = abt[j]; }
    }
    __syncthreads();

    float *my = slab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(D * (K + 5));
    const int d = threadIdx.x;
    if (d < D) {
        my[d] = red[0 * D + d] + red[2 * D + d] + red[4 * D + d] + red[6 * D + d];
        my[D + d] = red[1 * D + d] + red[3 * D + d] + red[5 * D + d] + red[7 * D + d];
        float w[K], dw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { w[k] = cw[d * K + k]; dw[k] = 0.f; }
        float dcb = 0.f, dba = 0.f, dbb = 0.f;
        const float ba = b2 ? b2[d] : 0.f, bb = b2 ? b2[D + d] : 0.f;
        const int own_off = K - 1 - pad_l;  // dc_lds row of frame t0
        for (int s = 0; s < CM_TT; ++s) {
            const int t = t0 + s;
            if (t >= Tn) break;
            // dW[k] += dc[t] * g[t + k - pad_l] ; dcb += dc[t]
            const float dct = dc_lds[(own_off + s) * D + d];
            dcb += dct;
            float dg = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                dw[k] += dct * g_lds[(s + k) * D + d];
                // dg[t] = sum_k w[k] * dc[t - k + pad_l]  -> dc_lds row (t - k + pad_l) - dc_row0 = s + (K-1) - k
                dg += w[k] * dc_lds[(s + (K - 1) - k) * D + d];
            }
            const T *p = y2 + ((size_t)b * Tn + t) * 2 * D;
            const float a = ld1(p + d) + ba, sg = sigmoidf_fast(ld1(p + D + d) + bb);
            const float da = dg * sg, db = dg * a * sg * (1.f - sg);
            T *q = dy2 + ((size_t)b * Tn + t) * 2 * D;
            st1(q + d, da);
            st1(q + D + d, db);
            dba += da;
            dbb += db;
        }
        my[2 * D + d] = dcb;
        my[3 * D + d] = dba;
        my[4 * D + d] = dbb;
#pragma unroll
        for (int k = 0; k < K; ++k) my[5 * D + d * K + k] = dw[k];
    }
}

__global__ __launch_bounds__(256) void convmod_colsum_kernel(const float *__restrict__ slab, float *__restrict__ out, int nparts,
                                                             int width) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < width) {
        int n = slice;
        for (; n + 48 < nparts; n += 64) {  // 4 independent loads in flight per lane
            s0 += slab[(size_t)n * width + col];
            s1 += slab[(size_t)(n + 16) * width + col];
            s2 += slab[(size_t)(n + 32) * width + col];
            s3 += slab[(size_t)(n + 48) * width + col];
        }
        for (; n < nparts; n += 16) s0 += slab[(size_t)n * width + col];
    }
    red[slice][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && col < width) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        out[col] = s;
    }
}

template <typename T, int K>
static void launch_fwd(const void *y2, const float *b2, const float *cw, const float *cb, const float *g, const float *be, void *z,
                       void *cs, float *mean, float *rstd, int B, int Tn, int D, int pad_l, float eps, float slope, hipStream_t st) {
    const size_t lds = (size_t)(2 * CM_TT + K - 1) * D * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)convmod_fwd_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    convmod_fwd_kernel<T, K><<<dim3(cdiv(Tn, CM_TT), B), 256, lds, st>>>((const T *)y2, b2, cw, cb, g, be, (T *)z, (T *)cs, mean, rstd, Tn, D, pad_l, eps, slope);
}

template <typename T, int K>
static void launch_bwd(const void *dz, const void *y2, const float *b2, const float *cw, const float *g, const float *be,
                       const void *cs, const float *mean, const float *rstd, void *dy2, float *slab, int B, int Tn, int D, int pad_l,
                       float slope, hipStream_t st) {
    const size_t lds = ((size_t)2 * (CM_TT + K - 1) * D + 8 * D) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)convmod_bwd_kernel<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    convmod_bwd_kernel<T, K><<<dim3(cdiv(Tn, CM_TT), B), 256, lds, st>>>((const T *)dz, (const T *)y2, b2, cw, g, be, (const T *)cs, mean, rstd, (T *)dy2, slab, Tn, D, pad_l, slope);
}

extern "C" {

/* y2 [B,T,2D] (bottleneck GEMM output WITHOUT its bias; b2 = that bias [2D] or NULL), conv_w [D,K] (the [D,1,K] parameter),
 * conv_b [D], gamma/beta [D]  ->  z [B,T,D] = LeakyReLU(LN(dwconv(GLU(y2+b2)))) ; c_save [B,T,D], mean/rstd [B*T] kept for bwd. */
int tsasr_convmod_fwd(const void *y2, const float *b2, const float *conv_w, const float *conv_b, const float *gamma,
                      const float *beta, void *z, void *c_save, float *mean, float *rstd, int B, int T, int D, int K, int causal,
                      float eps, float slope, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(y2 && conv_w && conv_b && gamma && beta && z && c_save && mean && rstd, "tsasr_convmod_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= CM_MAXD, "tsasr_convmod_fwd: D=%d must be a multiple of 8 and <= %d", D, CM_MAXD);
    TSASR_CHECK_ARG(K == 31 || K == 15 || K == 7 || K == 3, "tsasr_convmod_fwd: kernel size %d not instantiated (31, 15, 7, 3)", K);
    const int pad_l = causal ? K - 1 : (K - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
#define CM_F(TT, KK) launch_fwd<TT, KK>(y2, b2, conv_w, conv_b, gamma, beta, z, c_save, mean, rstd, B, T, D, pad_l, eps, slope, st)
#define CM_FK(TT) do { if (K == 31) CM_F(TT, 31); else if (K == 15) CM_F(TT, 15); else if (K == 7) CM_F(TT, 7); else CM_F(TT, 3); } while (0)
    if (io_dtype == TSASR_F32) CM_FK(float);
    else if (io_dtype == TSASR_BF16) CM_FK(bf16_t);
    else TSASR_CHECK_ARG(false, "tsasr_convmod_fwd: bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_convmod_fwd");
    return 0;
}

size_t tsasr_convmod_bwd_workspace_bytes(int B, int T, int D, int K) {
    return align_up((size_t)B * cdiv(T, CM_TT) * D * (K + 5) * sizeof(float), 256);
}

/* grads: dy2 [B,T,2D]; dparams fp32 packed [dgamma D | dbeta D | dconv_b D | db2 2D | dconv_w D*K] (OVERWRITTEN). */
int tsasr_convmod_bwd(const void *dz, const void *y2, const float *b2, const float *conv_w, const float *gamma, const float *beta,
                      const void *c_save, const float *mean, const float *rstd, void *dy2, float *dparams, int B, int T, int D,
                      int K, int causal, float slope, int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dz && y2 && conv_w && gamma && beta && c_save && mean && rstd && dy2 && dparams && workspace, "tsasr_convmod_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 8 == 0 && D <= CM_MAXD, "tsasr_convmod_bwd: bad D=%d", D);
    TSASR_CHECK_ARG(K == 31 || K == 15 || K == 7 || K == 3, "tsasr_convmod_bwd: kernel size %d not instantiated", K);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_convmod_bwd_workspace_bytes(B, T, D, K), "tsasr_convmod_bwd: workspace too small");
    const int pad_l = causal ? K - 1 : (K - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    float *slab = (float *)workspace;
#define CM_B(TT, KK) launch_bwd<TT, KK>(dz, y2, b2, conv_w, gamma, beta, c_save, mean, rstd, dy2, slab, B, T, D, pad_l, slope, st)
#define CM_BK(TT) do { if (K == 31) CM_B(TT, 31); else if (K == 15) CM_B(TT, 15); else if (K == 7) CM_B(TT, 7); else CM_B(TT, 3); } while (0)
    if (io_dtype == TSASR_F32) CM_BK(float);
    else if (io_dtype == TSASR_BF16) CM_BK(bf16_t);
    else TSASR_CHECK_ARG(false, "tsasr_convmod_bwd: bad io_dtype %d", io_dtype);
    const int width = D * (K + 5), nparts = B * cdiv(T, CM_TT);
    convmod_colsum_kernel<<<cdiv(width, 16), 256, 0, st>>>(slab, dparams, nparts, width);
    TSASR_CHECK_LAUNCH("tsasr_convmod_bwd");
    return 0;
}

}  // extern "C"
