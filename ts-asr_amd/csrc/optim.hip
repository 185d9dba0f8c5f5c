// Gradient-norm clipping + AdamW over the flat parameter arena, for gfx950.
//
// Replaces speechbrain/core.py:1082-1093: torch.nn.utils.clip_grad_norm_(modules.parameters(), max_grad_norm) ->
// optimizer.step() (torch.optim.AdamW, hparams conformer-t_scratch.yaml:267-271) -> zero_grad: ~190 tensors x several
// foreach passes plus a host sync for grad_norm.item(). Here: two launches over four flat fp32 buffers
//   (1) sumsq_partials: per-workgroup partial sums of g^2 (fixed partition -> deterministic);
//   (2) clip_adamw:     every workgroup re-reduces the partials (<= 1024 floats), derives the clip coefficient
//                       min(1, max_norm / (norm + 1e-6)) exactly as clip_grad_norm_, and applies decoupled weight decay,
//                       moment updates and the bias-corrected step in one read-modify-write pass (p, g, m, v: 7 x 4 B/elem).
// lr and the bias corrections come from a tiny DEVICE array so the step can sit inside a captured hipGraph while the
// Noam schedule keeps changing them from the host.
#include "common.h"

#define OPT_PARTS 1024

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float *__restrict__ g, float *__restrict__ part, long long n) {
    __shared__ float red[4];
    // a part is a multiple of 4 elements, so every part starts 16-byte aligned (with n / 1024 rounded up to an odd number three parts
    // out of four started misaligned and fell back to scalar loads: 82 us for 204 MB)
    const long long per = (((n + OPT_PARTS - 1) / OPT_PARTS) + 3) & ~3ll;
    const long long lo = min(n, (long long)blockIdx.x * per), hi = min(n, lo + per);
    const long long hi4 = lo + ((hi - lo) & ~3ll);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long i = lo + threadIdx.x * 4;
    for (; i + 3 * 1024 + 4 <= hi4; i += 4 * 1024) {     // four independent 16-byte loads in flight per thread
        const float4 a = *reinterpret_cast<const float4 *>(g + i), b = *reinterpret_cast<const float4 *>(g + i + 1024);
        const float4 c = *reinterpret_cast<const float4 *>(g + i + 2048), d = *reinterpret_cast<const float4 *>(g + i + 3072);
        s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    }
    for (; i + 4 <= hi4; i += 1024) {
        const float4 a = *reinterpret_cast<const float4 *>(g + i);
        s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    if (threadIdx.x == 0)
        for (long long k = hi4; k < hi; ++k) s0 += g[k] * g[k];
    float s = (s0 + s1) + (s2 + s3);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// hyper = {lr, 1 - beta1^t, 1 - beta2^t}
__global__ __launch_bounds__(256) void clip_adamw_kernel(float *__restrict__ p, bf16_t *__restrict__ p16, const float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v, const float *__restrict__ part,
                                                         const float *__restrict__ hyper, float *__restrict__ norm_out, float *__restrict__ skipped_out,
                                                         long long n, float beta1, float beta2, float eps, float wd, float max_norm) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < OPT_PARTS; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    // A non-finite gradient norm (overflow, or a poisoned activation: csrc/lstm.hip raises NaN when its inter-workgroup wait times out).
    // Reference behaviour (SB/core.py:1072-1093): the step is applied all the same - clip_grad_norm_'s factor max_norm / (norm + 1e-6),
    // clamped to at most 1, is NaN for a NaN norm and 0 for an infinite one, and AdamW then writes what that gives into every parameter and
    // moment; the non-finite LOSS was counted (check_gradients) and the run stops when nonfinite_patience is exhausted. That is what happens
    // here when skipped_out is NULL (`skip_nonfinite_step: False`, the default). With skipped_out (`skip_nonfinite_step: True`, a build
    // option) the step is skipped instead - parameters and moments untouched - and counted there.
    const bool finite = fabsf(norm) <= 3.0e38f;
    if (!finite && skipped_out) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *skipped_out += 1.f;   // skipped steps since the host last cleared it
        return;
    }
    float clip = 1.f;
    if (max_norm > 0.f) {
        const float c = max_norm / (norm + 1e-6f);
        clip = (c != c) ? c : fminf(1.f, c);        // torch.clamp(c, max=1): NaN stays NaN
    }
    const float lr = hyper[0], bc1 = hyper[1], bc2s = sqrtf(hyper[2]);
    const float step = lr / bc1, decay = 1.f - lr * wd;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 256 * 4) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            const float4 gg = *reinterpret_cast<const float4 *>(g + i);
            float *P = &pp.x, *M = &mm.x, *V = &vv.x;
            const float *G = &gg.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = G[k] * clip;
                M[k] = beta1 * M[k] + (1.f - beta1) * gk;
                V[k] = beta2 * V[k] + (1.f - beta2) * gk * gk;
                P[k] = P[k] * decay - step * M[k] / (sqrtf(V[k]) / bc2s + eps);
            }
            *reinterpret_cast<float4 *>(p + i) = pp;
            if (p16) {
                bf16x4 h4;
                h4[0] = (bf16_t)pp.x; h4[1] = (bf16_t)pp.y; h4[2] = (bf16_t)pp.z; h4[3] = (bf16_t)pp.w;
                *reinterpret_cast<bf16x4 *>(p16 + i) = h4;
            }
            *reinterpret_cast<float4 *>(m + i) = mm;
            *reinterpret_cast<float4 *>(v + i) = vv;
        } else {
            for (long long k = i; k < n; ++k) {
                const float gk = g[k] * clip;
                m[k] = beta1 * m[k] + (1.f - beta1) * gk;
                v[k] = beta2 * v[k] + (1.f - beta2) * gk * gk;
                p[k] = p[k] * decay - step * m[k] / (sqrtf(v[k]) / bc2s + eps);
                if (p16) p16[k] = (bf16_t)p[k];
            }
        }
    }
}

// dst[i][0..n[i]) += src[i][0..n[i])  for `count` fp32 vectors: ACC_SPLIT workgroups per vector, interleaved 256-element pieces (most vectors are
// biases and LayerNorm rows, a few are filter gradients of 10^5 elements: walked by ONE workgroup they were 576 dependent read-modify-write rounds
// at the very end of the step)
#define ACC_SPLIT 16
__global__ __launch_bounds__(256) void accumulate_many_kernel(const float *const *__restrict__ src, float *const *__restrict__ dst,
                                                              const int *__restrict__ n) {
    const float *s = src[blockIdx.x];
    float *d = dst[blockIdx.x];
    const int len = n[blockIdx.x];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < len; i += 256 * ACC_SPLIT) d[i] += s[i];
}

// dst_j[c][r] = src_j[r][c] for a list of 2-D bf16 matrices, one launch: transposed copies of the GEMM weights, refreshed after
// every optimizer step so that the input-gradient GEMMs (dx = dy . W) read W^T as a k-contiguous [K', N'] operand (the same
// fast path as the forward GEMM) instead of through transposing LDS reads (41 vs 22 us on the 8000 x 256 x 2048 FFN dgrad).
// jobs: int32 [njobs][5] = {src offset, dst offset (elements), rows, cols, first tile}; tile = 64 x 64 through LDS.
__global__ __launch_bounds__(256) void transpose_many_bf16_kernel(const unsigned short *__restrict__ src_base, unsigned short *__restrict__ dst_base,
                                                                  const int *__restrict__ jobs, int njobs) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = njobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {   // last job whose first tile <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid * 5 + 4] <= bid) lo = mid; else hi = mid - 1;
    }
    const int *jb = jobs + lo * 5;
    const int rows = jb[2], cols = jb[3], lt = bid - jb[4], tcn = (cols + 63) >> 6;
    const int r0 = (lt / tcn) * 64, c0 = (lt % tcn) * 64;
    const unsigned short *src = src_base + jb[0];
    unsigned short *dst = dst_base + jb[1];
    if (r0 + 64 <= rows && c0 + 64 <= cols && (cols & 7) == 0 && (rows & 7) == 0 && ((jb[0] | jb[1]) & 7) == 0) {
        // whole tile inside the matrix, 16-byte aligned rows (every GEMM weight of the model): two 16-byte loads and two 16-byte stores
        // per thread, all unconditional. The general form below moves 2 bytes per access behind per-element guards - 16 guarded loads =
        // 16 serialized round trips per workgroup (87 us for the step's 60 MB of weights: 1.4 TB/s)
        uint4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = threadIdx.x + 256 * i, row = q >> 3, c8 = (q & 7) * 8;
            v[i] = *reinterpret_cast<const uint4 *>(src + (size_t)(r0 + row) * cols + c0 + c8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = threadIdx.x + 256 * i, row = q >> 3, c8 = (q & 7) * 8;
            unsigned *t32 = reinterpret_cast<unsigned *>(&tile[row][c8]);      // row stride 132 B, c8 even: 4-byte aligned
            t32[0] = v[i].x; t32[1] = v[i].y; t32[2] = v[i].z; t32[3] = v[i].w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = threadIdx.x + 256 * i, c = q >> 3, r8 = (q & 7) * 8;
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tile[r8 + j][c];
            uint4 o;
            o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
            o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
            *reinterpret_cast<uint4 *>(dst + (size_t)(c0 + c) * rows + r0 + r8) = o;
        }
        return;
    }
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ry + 4 * i, c = c0 + cx;
        tile[ry + 4 * i][cx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ry + 4 * i, r = r0 + cx;
        if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[cx][ry + 4 * i];
    }
}

extern "C" {

/* Adds `count` small fp32 gradient vectors into their slots of the gradient arena with ONE launch (replaces one
 * AccumulateGrad add kernel per parameter: ~460 per step for LayerNorm weights, biases, conv filters, ...).
 * table: DEVICE memory laid out as [count src pointers][count dst pointers][count int32 lengths]. */
int tsasr_accumulate_many(const void *table, int count, void *stream) {
    TSASR_CHECK_ARG(table && count > 0, "tsasr_accumulate_many: bad arguments");
    const float *const *src = (const float *const *)table;
    float *const *dst = (float *const *)((const char *)table + (size_t)count * sizeof(void *));
    const int *n = (const int *)((const char *)table + (size_t)2 * count * sizeof(void *));
    accumulate_many_kernel<<<dim3(count, ACC_SPLIT), 256, 0, (hipStream_t)stream>>>(src, dst, n);
    TSASR_CHECK_LAUNCH("tsasr_accumulate_many");
    return 0;
}

size_t tsasr_clip_adamw_workspace_bytes(void) { return OPT_PARTS * sizeof(float); }

/* p, g, m, v: flat fp32 [n] (16-byte aligned); p_bf16: optional bf16 shadow of p (GEMM operand copy), rewritten in the same pass; hyper: DEVICE float[3] = {lr, 1-beta1^t, 1-beta2^t}; norm_out: device float
 * (may be NULL) = total L2 norm of g before clipping; skipped_out: device float or NULL. NULL (reference behaviour, SB/core.py:1072-1093): a
 * non-finite norm does not stop the update (clip factor NaN / 0 as torch.nn.utils.clip_grad_norm_ computes it). Not NULL: such a step is
 * skipped (parameters and moments untouched) and *skipped_out += 1. max_norm <= 0 disables clipping. g is read, not modified. */
int tsasr_clip_adamw_step(float *p, void *p_bf16, const float *g, float *m, float *v, const float *hyper, float *norm_out, float *skipped_out, long long n,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, void *workspace,
                          size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(p && g && m && v && hyper && workspace, "tsasr_clip_adamw_step: null pointer");
    TSASR_CHECK_ARG(n > 0 && workspace_bytes >= tsasr_clip_adamw_workspace_bytes(), "tsasr_clip_adamw_step: bad size / workspace");
    TSASR_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "tsasr_clip_adamw_step: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    sumsq_partials_kernel<<<OPT_PARTS, 256, 0, st>>>(g, part, n);
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    clip_adamw_kernel<<<(unsigned)blocks, 256, 0, st>>>(p, (bf16_t *)p_bf16, g, m, v, part, hyper, norm_out, skipped_out, n, beta1, beta2, eps, weight_decay, max_norm);
    TSASR_CHECK_LAUNCH("tsasr_clip_adamw_step");
    return 0;
}

/* Transposed bf16 copies of `njobs` row-major matrices in ONE launch. jobs: DEVICE int32 [njobs][5] = {src offset, dst offset
 * (elements from src_base / dst_base), rows, cols, index of the matrix' first 64x64 tile}; ntiles = total tile count. */
int tsasr_transpose_many_bf16(const void *src_base, void *dst_base, const void *jobs, int njobs, int ntiles, void *stream) {
    TSASR_CHECK_ARG(src_base && dst_base && jobs && njobs > 0 && ntiles > 0, "tsasr_transpose_many_bf16: bad arguments");
    transpose_many_bf16_kernel<<<ntiles, 256, 0, (hipStream_t)stream>>>((const unsigned short *)src_base, (unsigned short *)dst_base, (const int *)jobs, njobs);
    TSASR_CHECK_LAUNCH("tsasr_transpose_many_bf16");
    return 0;
}

}  // extern "C"
