// Convolutional front-end (2 x stride-2 3x3 Conv2d + 1x1 stride-2 residual conv) data movement for gfx950.
//
// Replaces, for speechbrain/lobes/models/convolution.py:103-266 (ConvolutionFrontEnd / ConvBlock) and
// speechbrain/nnet/CNN.py:629-711 (Conv2d.forward with 'same' = reflect padding, or 'causal' = (k-1, 0) zero padding on
// time and (1, 1) zero padding on frequency), the pad kernels + MIOpen convolutions the reference runs
// (on ROCm: reflection_pad2d_backward 2.4 ms, naive/CK bwd-weight kernels up to 66 ms per call - profiles/r01_*).
//
// Layout is the reference's own [B, T, F, C] (channels-last): a 3x3 tap is a contiguous C-vector.
//   block 1 (C_in = 1): direct kernel, both branches (3x3 and 1x1) at once, thread = 8 output channels of one position,
//                       filters in registers; backward gives the filter gradients (the input is the features: no dx).
//   block 2 (C_in = 128): the contraction is a plain [P, 9C] x [9C, C_out] GEMM once the taps are gathered, so it goes to
//                       the library GEMM; this file does the gather (im2col, with the padding rule folded into the
//                       source index) and the deterministic inverse gather-sum (col2im) for the input gradient -
//                       every lane moves 16 bytes, rows are 256-byte contiguous C-vectors.
#include "common.h"

// source index along one axis for output index o, tap k (0..2), stride 2
//   mode 0 ('same'):  i = 2o + k - 1, reflected at both ends      (CNN.py:678-711, F.pad(mode="reflect"))
//   mode 1 (causal, time axis): i = 2o + k - 2, zero (-1) if < 0  (CNN.py:649-657)
//   mode 2 (zero pad 1/1, frequency axis of the causal front-end): i = 2o + k - 1, -1 outside
__device__ __forceinline__ int src_index(int o, int k, int n, int mode) {
    if (mode == 1) { const int i = 2 * o + k - 2; return i < 0 ? -1 : i; }
    int i = 2 * o + k - 1;
    if (mode == 0) { if (i < 0) i = -i; if (i >= n) i = 2 * (n - 1) - i; return i; }
    return (i < 0 || i >= n) ? -1 : i;
}
__host__ __device__ static inline int out_len(int n) { return (n - 1) / 2 + 1; }

// ---------------------------------------------------------------------------------------------------
// block 1: C_in = 1
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fe_c1_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w1 /*[C][3(kf)][3(kt)]*/,
                                                        const float *__restrict__ b1, const float *__restrict__ w2 /*[C]*/,
                                                        const float *__restrict__ b2, T *__restrict__ y1, T *__restrict__ y2,
                                                        int B, int Tn, int F, int C, int tmode, int fmode) {
    const int To = out_len(Tn), Fo = out_len(F);
    const int cg = threadIdx.x % (C / 8), pl = threadIdx.x / (C / 8), ppb = 256 / (C / 8);
    float w[8][9], bb1[8], ww2[8], bb2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
#pragma unroll
        for (int k = 0; k < 9; ++k) w[j][k] = w1[c * 9 + k];
        bb1[j] = b1[c]; ww2[j] = w2[c]; bb2[j] = b2[c];
    }
    const long long P = (long long)B * To * Fo;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < P; p += (long long)gridDim.x * ppb) {
        const int fo = p % Fo, to = (p / Fo) % To, b = p / ((long long)Fo * To);
        float a[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int ti = src_index(to, kt, Tn, tmode);
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int fi = src_index(fo, kf, F, fmode);
                // always issued (clamped address), masked afterwards: a conditional load is waited for where it stands
                const float xv = ld1(x + ((size_t)b * Tn + max(ti, 0)) * F + max(fi, 0));
                a[kf * 3 + kt] = (ti >= 0 && fi >= 0) ? xv : 0.f;  // weight index = kf*3 + kt
            }
        }
        const float centre = ld1(x + ((size_t)b * Tn + 2 * to) * F + 2 * fo);  // 1x1 stride-2 conv: no padding
        float o1[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = bb1[j];
#pragma unroll
            for (int k = 0; k < 9; ++k) s += w[j][k] * a[k];
            o1[j] = s;
            o2[j] = bb2[j] + ww2[j] * centre;
        }
        st8(y1 + p * C + cg * 8, o1);
        st8(y2 + p * C + cg * 8, o2);
    }
}

// slab row per workgroup: [dw1 C*9][db1 C][dw2 C][db2 C]
template <typename T>
__global__ __launch_bounds__(256) void fe_c1_bwd_kernel(const T *__restrict__ x, const T *__restrict__ dy1, const T *__restrict__ dy2,
                                                        float *__restrict__ slab, int B, int Tn, int F, int C, int tmode, int fmode) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][C*12]
    const int To = out_len(Tn), Fo = out_len(F);
    const int cg = threadIdx.x % (C / 8), pl = threadIdx.x / (C / 8), ppb = 256 / (C / 8);
    float dw[8][9], db1[8], dw2[8], db2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dw[j][k] = 0.f;
        db1[j] = dw2[j] = db2[j] = 0.f;
    }
    const long long P = (long long)B * To * Fo;
    // The bias sums go through the same packed FMA form as the filter taps (x 1.0, opaque to the compiler). Written as plain adds
    // they compiled to v_pk_add_f32 with swapped op_sel halves (op_sel:[0,1] op_sel_hi:[1,0]) - the only such instruction in the
    // library - and exactly those sums (even channels of db1) came out different in the low bits from run to run whenever a second
    // hardware queue had kernels in flight (hipGraph replay with the forked speaker branch): an instruction-timing hazard, not a
    // data race (found by hashing every parameter gradient of identical replays, tests/test_recipe_gpu.py).
    float one = 1.f;
    asm volatile("" : "+v"(one));
    for (long long p = (long long)blockIdx.x * ppb + pl; p < P; p += (long long)gridDim.x * ppb) {
        const int fo = p % Fo, to = (p / Fo) % To, b = p / ((long long)Fo * To);
        float a[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int ti = src_index(to, kt, Tn, tmode);
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int fi = src_index(fo, kf, F, fmode);
                const float xv = ld1(x + ((size_t)b * Tn + max(ti, 0)) * F + max(fi, 0));
                a[kf * 3 + kt] = (ti >= 0 && fi >= 0) ? xv : 0.f;
            }
        }
        const float centre = ld1(x + ((size_t)b * Tn + 2 * to) * F + 2 * fo);
        float g1[8], g2[8];
        ld8(dy1 + p * C + cg * 8, g1);
        ld8(dy2 + p * C + cg * 8, g2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) dw[j][k] += g1[j] * a[k];
            db1[j] = __builtin_fmaf(g1[j], one, db1[j]);   // (see `one` above)
            dw2[j] += g2[j] * centre;
            db2[j] = __builtin_fmaf(g2[j], one, db2[j]);
        }
    }
    // lanes of a wave that share a channel group differ in the position bits: fold them with xor-shuffles first, so that the
    // workgroup needs 4 x C*12 floats of LDS (12 KB at C = 64) instead of ppb x C*12 (96 KB = one workgroup per CU, every
    // loop iteration's memory round trip exposed) and many workgroups share a CU
    const int W = C * 12, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cgs = C / 8;
    for (int off = cgs; off < 32; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) dw[j][k] += __shfl_xor(dw[j][k], off, 64);
            db1[j] += __shfl_xor(db1[j], off, 64);
            dw2[j] += __shfl_xor(dw2[j], off, 64);
            db2[j] += __shfl_xor(db2[j], off, 64);
        }
    }
    if (cgs <= 32)
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // the last fold (lane ^ 32) is a v_permlane32_swap, not a trip through the LDS crossbar
#pragma unroll
        for (int k = 0; k < 9; ++k) dw[j][k] += other_half(dw[j][k]);
        db1[j] += other_half(db1[j]);
        dw2[j] += other_half(dw2[j]);
        db2[j] += other_half(db2[j]);
    }
    if (lane < cgs) {
        float *mine = red + wave * W;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
#pragma unroll
            for (int k = 0; k < 9; ++k) mine[c * 9 + k] = dw[j][k];
            mine[C * 9 + c] = db1[j];
            mine[C * 10 + c] = dw2[j];
            mine[C * 11 + c] = db2[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) slab[(size_t)blockIdx.x * W + i] = (red[i] + red[W + i]) + (red[2 * W + i] + red[3 * W + i]);
}


// ---------------------------------------------------------------------------------------------------
// block 2: gather / inverse gather of 3x3 stride-2 taps, C-vectors of 16-byte chunks
// A[p][(kt*3 + kf)*C + c] = x[b][src(to,kt)][src(fo,kf)][c]   (0 where the source is padding)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fe_im2col_kernel(const T *__restrict__ x, T *__restrict__ A, int B, int Tn, int F, int C,
                                                        int tmode, int fmode) {
    constexpr int VE = 16 / sizeof(T);
    const int To = out_len(Tn), Fo = out_len(F), cv = C / VE;
    const long long total = (long long)B * To * Fo * 9 * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        const int tap = (int)((i / cv) % 9);
        const long long p = i / ((long long)cv * 9);
        const int fo = p % Fo, to = (p / Fo) % To, b = p / ((long long)Fo * To);
        const int ti = src_index(to, tap / 3, Tn, tmode), fi = src_index(fo, tap % 3, F, fmode);
        uint4 v = *reinterpret_cast<const uint4 *>(x + (((size_t)b * Tn + max(ti, 0)) * F + max(fi, 0)) * C + c * VE);   // always issued
        if (ti < 0 || fi < 0) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(A + (p * 9 + tap) * C + c * VE) = v;
    }
}

// dx[b][t][f][:] = sum over (to,kt),(fo,kf) with src == (t,f) of dA[p][tap][:]  (+ dR[p][:] at t = 2to, f = 2fo)
template <typename T>
__global__ __launch_bounds__(256) void fe_col2im_kernel(const T *__restrict__ dA, const T *__restrict__ dR, T *__restrict__ dx, int B,
                                                        int Tn, int F, int C, int tmode, int fmode) {
    constexpr int VE = 16 / sizeof(T);
    const int To = out_len(Tn), Fo = out_len(F), cv = C / VE;
    const long long total = (long long)B * Tn * F * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        const long long q = i / cv;
        const int f = q % F, t = (q / F) % Tn, b = q / ((long long)F * Tn);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
        for (int to = max(0, t / 2 - 1); to <= min(To - 1, t / 2 + 1); ++to)
            for (int kt = 0; kt < 3; ++kt) {
                if (src_index(to, kt, Tn, tmode) != t) continue;
                for (int fo = max(0, f / 2 - 1); fo <= min(Fo - 1, f / 2 + 1); ++fo)
                    for (int kf = 0; kf < 3; ++kf) {
                        if (src_index(fo, kf, F, fmode) != f) continue;
                        const long long p = ((long long)b * To + to) * Fo + fo;
                        float v[VE];
                        if constexpr (VE == 8) ld8(reinterpret_cast<const bf16_t *>(dA) + (p * 9 + kt * 3 + kf) * C + c * VE, *reinterpret_cast<float(*)[8]>(&v[0]));
                        else {
                            const float4 u = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(dA) + (p * 9 + kt * 3 + kf) * C + c * VE);
                            v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
                        }
#pragma unroll
                        for (int j = 0; j < VE; ++j) acc[j] += v[j];
                    }
            }
        if (dR && (t % 2 == 0) && (f % 2 == 0) && t / 2 < To && f / 2 < Fo) {
            const long long p = ((long long)b * To + t / 2) * Fo + f / 2;
#pragma unroll
            for (int j = 0; j < VE; ++j) acc[j] += ld1(dR + p * C + c * VE + j);
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) st1(dx + q * C + c * VE + j, acc[j]);
    }
}

static unsigned grid_for(long long work_items) {
    long long b = (work_items + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}
static int pad_modes(int causal, int *tmode, int *fmode) {
    *tmode = causal ? 1 : 0;
    *fmode = causal ? 2 : 0;
    return 0;
}

extern "C" {

int tsasr_frontend_out_len(int n) { return out_len(n); }

/* Block 1 (C_in = 1). x [B,T,F]; w1 [C,1,3,3] as stored by the reference (kernel axes (F,T)); w2 [C,1,1,1];
 * y1 = conv3x3_s2(x) + b1, y2 = conv1x1_s2(x) + b2, both [B,T',F',C]. C % 8 == 0, C <= 2048. */
int tsasr_frontend_c1_fwd(const void *x, const float *w1, const float *b1, const float *w2, const float *b2, void *y1, void *y2,
                          int B, int T, int F, int C, int causal, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && w1 && b1 && w2 && b2 && y1 && y2, "tsasr_frontend_c1_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && C % 8 == 0 && C >= 8 && C <= 2048 && 256 % (C / 8) == 0, "tsasr_frontend_c1_fwd: bad shape (T=%d F=%d C=%d)", T, F, C);
    int tm, fm;
    pad_modes(causal, &tm, &fm);
    const long long P = (long long)B * out_len(T) * out_len(F);
    // few, long-lived workgroups: a thread keeps its 8 x 12 filter values in registers and walks ~10 output positions with them
    const unsigned grid = min(grid_for(P * (C / 8)), 2048u);
    if (io_dtype == TSASR_F32) fe_c1_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float *)x, w1, b1, w2, b2, (float *)y1, (float *)y2, B, T, F, C, tm, fm);
    else if (io_dtype == TSASR_BF16) fe_c1_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t *)x, w1, b1, w2, b2, (bf16_t *)y1, (bf16_t *)y2, B, T, F, C, tm, fm);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_frontend_c1_fwd");
    return 0;
}

#define FE_C1_BWD_WGS 2048
size_t tsasr_frontend_c1_bwd_workspace_bytes(int C) { return align_up((size_t)FE_C1_BWD_WGS * C * 12 * sizeof(float), 256); }

/* dparams fp32 packed [dw1 C*9 | db1 C | dw2 C | db2 C], overwritten. */
int tsasr_frontend_c1_bwd(const void *x, const void *dy1, const void *dy2, float *dparams, int B, int T, int F, int C, int causal,
                          int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(x && dy1 && dy2 && dparams && workspace, "tsasr_frontend_c1_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && C % 8 == 0 && C >= 8 && C <= 256 && 256 % (C / 8) == 0, "tsasr_frontend_c1_bwd: bad shape");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_frontend_c1_bwd_workspace_bytes(C), "tsasr_frontend_c1_bwd: workspace too small");
    int tm, fm;
    pad_modes(causal, &tm, &fm);
    const size_t lds = (size_t)4 * C * 12 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    float *slab = (float *)workspace;
    if (io_dtype == TSASR_F32) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fe_c1_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        fe_c1_bwd_kernel<float><<<FE_C1_BWD_WGS, 256, lds, st>>>((const float *)x, (const float *)dy1, (const float *)dy2, slab, B, T, F, C, tm, fm);
    } else if (io_dtype == TSASR_BF16) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fe_c1_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        fe_c1_bwd_kernel<bf16_t><<<FE_C1_BWD_WGS, 256, lds, st>>>((const bf16_t *)x, (const bf16_t *)dy1, (const bf16_t *)dy2, slab, B, T, F, C, tm, fm);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    tsasr_reduce_submit(slab, dparams, C * 12, FE_C1_BWD_WGS, C * 12, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_frontend_c1_bwd");
    return 0;
}

/* A [B*T'*F', 9*C] <- taps of x [B,T,F,C]; column order (kt, kf, c). C * sizeof(elem) % 16 == 0. */
int tsasr_frontend_im2col(const void *x, void *A, int B, int T, int F, int C, int causal, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(x && A, "tsasr_frontend_im2col: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && C > 0, "tsasr_frontend_im2col: bad shape");
    int tm, fm;
    pad_modes(causal, &tm, &fm);
    const long long P = (long long)B * out_len(T) * out_len(F);
    if (io_dtype == TSASR_F32) {
        TSASR_CHECK_ARG(C % 4 == 0, "tsasr_frontend_im2col: C=%d must be a multiple of 4", C);
        fe_im2col_kernel<float><<<grid_for(P * 9 * (C / 4)), 256, 0, (hipStream_t)stream>>>((const float *)x, (float *)A, B, T, F, C, tm, fm);
    } else if (io_dtype == TSASR_BF16) {
        TSASR_CHECK_ARG(C % 8 == 0, "tsasr_frontend_im2col: C=%d must be a multiple of 8", C);
        fe_im2col_kernel<bf16_t><<<grid_for(P * 9 * (C / 8)), 256, 0, (hipStream_t)stream>>>((const bf16_t *)x, (bf16_t *)A, B, T, F, C, tm, fm);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_frontend_im2col");
    return 0;
}

/* dx [B,T,F,C] <- dA [P, 9*C] (+ dR [P, C] from the 1x1 stride-2 branch, may be NULL). */
int tsasr_frontend_col2im(const void *dA, const void *dR, void *dx, int B, int T, int F, int C, int causal, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(dA && dx, "tsasr_frontend_col2im: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && C > 0, "tsasr_frontend_col2im: bad shape");
    int tm, fm;
    pad_modes(causal, &tm, &fm);
    const long long Q = (long long)B * T * F;
    if (io_dtype == TSASR_F32) {
        TSASR_CHECK_ARG(C % 4 == 0, "tsasr_frontend_col2im: C=%d must be a multiple of 4", C);
        fe_col2im_kernel<float><<<grid_for(Q * (C / 4)), 256, 0, (hipStream_t)stream>>>((const float *)dA, (const float *)dR, (float *)dx, B, T, F, C, tm, fm);
    } else if (io_dtype == TSASR_BF16) {
        TSASR_CHECK_ARG(C % 8 == 0, "tsasr_frontend_col2im: C=%d must be a multiple of 8", C);
        fe_col2im_kernel<bf16_t><<<grid_for(Q * (C / 8)), 256, 0, (hipStream_t)stream>>>((const bf16_t *)dA, (const bf16_t *)dR, (bf16_t *)dx, B, T, F, C, tm, fm);
    } else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_frontend_col2im");
    return 0;
}

}  // extern "C"
