// RNN-T joint + head in EXACT fp32 arithmetic (no matrix cores, no bf16 rounding): the `compute_dtype: fp32` parity mode of
// csrc/rnnt.hip's joint_fwd / joint_bwd (which round h = lrelu(enc + dec), the head matrix and dlogits to bf16 for their MFMA
// contractions whatever the storage type).
//
// Replaces, for fp32 activations: SB/nnet/transducer/transducer_joint.py:73-95 (joint "sum" + LeakyReLU) + SB/nnet/linear.py:64-78 (head),
// call site train_librispeechmix_scratch.py:132-135, and their autograd backward. The [B,T,U1,J] joint tensor is still never built:
//   forward : one workgroup per (b, t): head matrix [V][J] in LDS, thread (u, v) sums lrelu(enc[k] + dec[u][k]) * W[v][k] over k;
//   backward: X - workgroup (b, 8 lattice columns) walks t: thread k keeps W[:, k] in registers, forms
//                 dh = lrelu'(enc + dec) * (dlogits[u,:] . W[:, k]) and accumulates ddec[u][k] and dW[:, k] (slab per workgroup);
//             Y - workgroup (b, t) walks u: denc[k] = sum_u dh.
// Deterministic (fixed summation orders; the slabs go through csrc/reduce.hip). VALU speed: a checker-grade path.
#include "common.h"

namespace {

constexpr int VMAX = 32;       // vocabulary rows <= 32 (logits rows are 32 floats)
constexpr int UT = 8;          // lattice columns per pass
constexpr int KPT = 3;         // joint dims per thread in the backward (J <= 768)

__global__ __launch_bounds__(256) void joint_f32_fwd_kernel(const float *__restrict__ enc, const float *__restrict__ dec, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ logits, int T, int U1, int J, int V,
                                                            int ldl, float slope) {
    extern __shared__ float sm[];
    float *Ws = sm;                        // [VMAX][J + 1]
    float *es = Ws + VMAX * (J + 1);       // [J]
    float *ds = es + J;                    // [UT][J]
    const int b = blockIdx.y, t = blockIdx.x;
    for (int e = threadIdx.x; e < VMAX * J; e += 256) {
        const int v = e / J, k = e % J;
        Ws[v * (J + 1) + k] = v < V ? W[(long long)v * J + k] : 0.f;
    }
    for (int k = threadIdx.x; k < J; k += 256) es[k] = enc[((long long)b * T + t) * J + k];
    const int v = threadIdx.x & 31, ul = threadIdx.x >> 5;
    const float bv = (bias && v < V) ? bias[v] : 0.f;
    for (int u0 = 0; u0 < U1; u0 += UT) {
        __syncthreads();
        for (int e = threadIdx.x; e < UT * J; e += 256) {
            const int uu = e / J, k = e % J;
            ds[e] = u0 + uu < U1 ? dec[((long long)b * U1 + u0 + uu) * J + k] : 0.f;
        }
        __syncthreads();
        const int u = u0 + ul;
        float acc = 0.f;
        for (int k = 0; k < J; ++k) {
            const float z = es[k] + ds[ul * J + k];
            acc += lrelu(z, slope) * Ws[v * (J + 1) + k];
        }
        if (u < U1 && v < ldl) logits[(((long long)b * T + t) * U1 + u) * ldl + v] = v < V ? acc + bv : 0.f;
    }
}

// X: ddec + head-gradient slabs. grid (ceil(U1 / UT), B)
__global__ __launch_bounds__(256) void joint_f32_bwd_x_kernel(const float *__restrict__ dl, const float *__restrict__ enc, const float *__restrict__ dec,
                                                              const float *__restrict__ W, float *__restrict__ ddec, float *__restrict__ slab_w,
                                                              float *__restrict__ slab_b, const int *__restrict__ tlen, const int *__restrict__ ulen,
                                                              int T, int U1, int J, int V, int ldl, float slope) {
    __shared__ float g[UT][VMAX];
    const int b = blockIdx.y, u0 = blockIdx.x * UT;
    const int tl = tlen ? min(tlen[b], T) : T, ul = ulen ? min(ulen[b] + 1, U1) : U1;     // dlogits are zero outside the utterance's lattice
    float w[KPT][VMAX], dw[KPT][VMAX], dd[KPT][UT], dcv[KPT][UT], db = 0.f;
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        const int k = threadIdx.x + 256 * c;
#pragma unroll
        for (int v = 0; v < VMAX; ++v) {
            w[c][v] = (k < J && v < V) ? W[(long long)v * J + k] : 0.f;
            dw[c][v] = 0.f;
        }
#pragma unroll
        for (int uu = 0; uu < UT; ++uu) {
            dd[c][uu] = 0.f;
            dcv[c][uu] = (k < J && u0 + uu < U1) ? dec[((long long)b * U1 + u0 + uu) * J + k] : 0.f;
        }
    }
    for (int t = 0; t < tl; ++t) {
        __syncthreads();
        {
            const int uu = threadIdx.x >> 5, v = threadIdx.x & 31;
            const int u = u0 + uu;
            g[uu][v] = (u < ul && v < V) ? dl[(((long long)b * T + t) * U1 + u) * ldl + v] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < VMAX) {
            float s = 0.f;
#pragma unroll
            for (int uu = 0; uu < UT; ++uu) s += g[uu][threadIdx.x];
            db += s;
        }
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            const int k = threadIdx.x + 256 * c;
            if (k >= J) continue;
            const float e = enc[((long long)b * T + t) * J + k];
#pragma unroll
            for (int uu = 0; uu < UT; ++uu) {
                const float z = e + dcv[c][uu];
                const float h = lrelu(z, slope);
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < VMAX; ++v) {
                    const float gv = g[uu][v];
                    s += gv * w[c][v];
                    dw[c][v] += gv * h;
                }
                dd[c][uu] += z > 0.f ? s : s * slope;
            }
        }
    }
    const long long slab = (long long)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        const int k = threadIdx.x + 256 * c;
        if (k >= J) continue;
#pragma unroll
        for (int v = 0; v < VMAX; ++v) slab_w[(slab * VMAX + v) * J + k] = dw[c][v];
#pragma unroll
        for (int uu = 0; uu < UT; ++uu)
            if (u0 + uu < U1) ddec[((long long)b * U1 + u0 + uu) * J + k] = dd[c][uu];
    }
    if (threadIdx.x < VMAX) slab_b[slab * VMAX + threadIdx.x] = db;
}

// Y: denc. grid (T, B)
__global__ __launch_bounds__(256) void joint_f32_bwd_y_kernel(const float *__restrict__ dl, const float *__restrict__ enc, const float *__restrict__ dec,
                                                              const float *__restrict__ W, float *__restrict__ denc, const int *__restrict__ tlen,
                                                              const int *__restrict__ ulen, int T, int U1, int J, int V, int ldl, float slope) {
    __shared__ float g[UT][VMAX];
    const int b = blockIdx.y, t = blockIdx.x;
    const int tl = tlen ? min(tlen[b], T) : T, ul = ulen ? min(ulen[b] + 1, U1) : U1;
    float w[KPT][VMAX], e[KPT], acc[KPT];
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        const int k = threadIdx.x + 256 * c;
#pragma unroll
        for (int v = 0; v < VMAX; ++v) w[c][v] = (k < J && v < V) ? W[(long long)v * J + k] : 0.f;
        e[c] = k < J ? enc[((long long)b * T + t) * J + k] : 0.f;
        acc[c] = 0.f;
    }
    if (t < tl) {
        for (int u0 = 0; u0 < ul; u0 += UT) {
            __syncthreads();
            {
                const int uu = threadIdx.x >> 5, v = threadIdx.x & 31;
                const int u = u0 + uu;
                g[uu][v] = (u < ul && v < V) ? dl[(((long long)b * T + t) * U1 + u) * ldl + v] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < KPT; ++c) {
                const int k = threadIdx.x + 256 * c;
                if (k >= J) continue;
                for (int uu = 0; uu < UT; ++uu) {
                    if (u0 + uu >= ul) break;
                    const float z = e[c] + dec[((long long)b * U1 + u0 + uu) * J + k];
                    float s = 0.f;
#pragma unroll
                    for (int v = 0; v < VMAX; ++v) s += g[uu][v] * w[c][v];
                    acc[c] += z > 0.f ? s : s * slope;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        const int k = threadIdx.x + 256 * c;
        if (k < J) denc[((long long)b * T + t) * J + k] = acc[c];
    }
}

}  // namespace

extern "C" {

/* logits[b,t,u,v] = bias[v] + sum_k W[v,k] * lrelu(enc[b,t,k] + dec[b,u,k]) with every operand, product and sum in fp32. enc [B,T,J],
 * dec [B,U1,J], W [V,J], logits rows padded to ldl floats (columns >= V written as 0). V <= 32, J <= 768. */
int tsasr_joint_f32_fwd(const float *enc, const float *dec, const float *W, const float *bias, float *logits, int B, int T, int U1, int J, int V,
                        int ldl, float slope, void *stream) {
    TSASR_CHECK_ARG(enc && dec && W && logits, "tsasr_joint_f32_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && J > 0 && J <= 256 * KPT && V > 0 && V <= VMAX && ldl >= V && ldl <= VMAX,
                    "tsasr_joint_f32_fwd: unsupported sizes (J=%d <= %d, V=%d <= %d, ldl=%d)", J, 256 * KPT, V, VMAX, ldl);
    const size_t lds = ((size_t)VMAX * (J + 1) + J + (size_t)UT * J) * sizeof(float);
    TSASR_CHECK_ARG(lds <= 160 * 1024, "tsasr_joint_f32_fwd: J=%d needs %zu bytes of LDS", J, lds);
    (void)hipFuncSetAttribute((const void *)joint_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    joint_f32_fwd_kernel<<<dim3(T, B), 256, lds, (hipStream_t)stream>>>(enc, dec, W, bias, logits, T, U1, J, V, ldl, slope);
    TSASR_CHECK_LAUNCH("tsasr_joint_f32_fwd");
    return 0;
}

size_t tsasr_joint_f32_bwd_workspace_bytes(int B, int U1, int J) {
    const size_t nslab = (size_t)B * cdiv(U1, UT);
    return align_up(nslab * VMAX * J * sizeof(float), 256) + align_up(nslab * VMAX * sizeof(float), 256);
}

/* denc [B,T,J], ddec [B,U1,J] fully written; dW [V,J], dbias [V] through the deferrable batched reduction. tlen / ulen (may be NULL):
 * dlogits outside frame < tlen[b], column <= ulen[b] are taken as zero (the loss writes zeros there). */
int tsasr_joint_f32_bwd(const float *dlogits, const float *enc, const float *dec, const float *W, float *denc, float *ddec, float *dW,
                        float *dbias, const int32_t *tlen, const int32_t *ulen, int B, int T, int U1, int J, int V, int ldl, float slope,
                        void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dlogits && enc && dec && W && denc && ddec && dW && dbias && workspace, "tsasr_joint_f32_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && J > 0 && J <= 256 * KPT && V > 0 && V <= VMAX && ldl >= V && ldl <= VMAX,
                    "tsasr_joint_f32_bwd: unsupported sizes (J=%d V=%d ldl=%d)", J, V, ldl);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_joint_f32_bwd_workspace_bytes(B, U1, J), "tsasr_joint_f32_bwd: workspace too small");
    const int nut = cdiv(U1, UT);
    const size_t nslab = (size_t)B * nut;
    float *slab_w = (float *)workspace, *slab_b = (float *)((char *)workspace + align_up(nslab * VMAX * J * sizeof(float), 256));
    hipStream_t st = (hipStream_t)stream;
    joint_f32_bwd_x_kernel<<<dim3(nut, B), 256, 0, st>>>(dlogits, enc, dec, W, ddec, slab_w, slab_b, tlen, ulen, T, U1, J, V, ldl, slope);
    joint_f32_bwd_y_kernel<<<dim3(T, B), 256, 0, st>>>(dlogits, enc, dec, W, denc, tlen, ulen, T, U1, J, V, ldl, slope);
    TSASR_CHECK_LAUNCH("tsasr_joint_f32_bwd");
    tsasr_reduce_submit(slab_w, dW, (long long)VMAX * J, (int)nslab, V * J, 0, st);
    tsasr_reduce_submit(slab_b, dbias, VMAX, (int)nslab, V, 0, st);
    return 0;
}

}  // extern "C"
