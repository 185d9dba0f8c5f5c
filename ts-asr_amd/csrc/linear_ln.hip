// Output projection + residual tail + LayerNorm in ONE launch, bf16, d_model = 256:
//     x = A[M,256] . W[256,256]^T          (rounded to bf16, as the stand-alone GEMM stores it)
//     s = res + alpha * timemask(dropout_p(x + bias)) ;   y = LayerNorm(s) * gamma + beta
//
// Replaces, at the two seams of a Conformer layer whose GEMM has K = 256 (the attention's out_proj and the convolution module's second
// point-wise convolution: SB/nnet/attention.py:549-553 `out_proj`, SB/lobes/models/transformer/Conformer.py:76-98 `after_conv`, then
// Conformer.py:243-259 `x = x + skip` / `x + self.convolution_module(..)` and the LayerNorm that reads it, :194-217), the pair
// gemm_bf16_ring_kernel<128,64> (5.3 us at M = 8000) + add_layernorm_fwd_kernel (6.7 us): both are a launch and one round trip each,
// and the 4 MB between them goes out to L2 / the Infinity Cache and comes back. Same arithmetic in the same order, so (s, y, mean, rstd)
// are bit-identical to the pair (tests/test_blocks_gpu.py): the accumulators take the same v_mfma_f32_32x32x16_bf16 steps over k, are rounded to
// bf16, and the row pass below is add_layernorm_fwd_kernel<bf16_t, 1, true>'s (two rows per wave, lane = 8 columns).
//
// Workgroup = 256 threads = 4 waves, one 32-row x 256-column tile (250 workgroups at M = 8000): a row of the output needs all 256
// columns, so a tile is a full row panel and every workgroup stages ALL of W (128 KiB, from L2) - affordable at K = 256 only (at K = 2048,
// the FFN's down-projection, it would be 1 MiB per workgroup: 250 MiB through the CUs' load paths). Everything is requested at once by
// LDS-DMA as four k-tiles of [32 + 256 rows][64 k] (csrc/gemm_big.hip's layout: 128-byte rows, XOR swizzle on the source chunk), the
// MFMAs of k-tile kt start when its 36 KiB have landed (counted s_waitcnt, one s_barrier per k-tile) while the later tiles stream in;
// wave w owns columns [64 w, 64 w + 64). The fp32 accumulators then go through LDS (the space of k-tile 0) and leave row-major.
#include "common.h"

#define LL_BM 32
#define LL_D 256
#define LL_ROW 128                      // bytes per LDS row (64 bf16 of k)
#define LL_SLOT ((LL_BM + LL_D) * LL_ROW)   // one k-tile: 36 KiB
#define LL_PCS (LL_SLOT / 1024 / 4)     // 1 KiB DMA pieces per wave and k-tile: 9
#define LL_LDF (LL_D + 4)               // fp32 epilogue tile row stride

namespace {

__device__ __forceinline__ void ll_dma16(const bf16_t *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void linear_add_layernorm_fwd_kernel(const bf16_t *__restrict__ A, long long lda, const bf16_t *__restrict__ W,
                                                                      long long ldw, const float *__restrict__ bias,
                                                                      const bf16_t *__restrict__ res, bf16_t *__restrict__ s_out,
                                                                      bf16_t *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta, long long M,
                                                                      float alpha, float p, unsigned long long seed,
                                                                      const unsigned long long *__restrict__ seed_dev,
                                                                      const int32_t *__restrict__ valid_lens, int Trows, float eps) {
    constexpr int D = LL_D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * LL_BM;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // ---- operands of the row pass, requested first (older than every DMA: they have landed when the first k-tile has) ----------------
    // thread = (wave, half hf, lane l of the half): rows 8 q + 2 wave + hf (q = 0 .. 3), columns [8 l, 8 l + 8) - the mapping of
    // add_layernorm_fwd_kernel<bf16_t, 1, true> (two rows per wave)
    const int l = lane & 31, hf = lane >> 5, c = l * 8;
    const bool has_bias = bias != nullptr, has_vl = valid_lens != nullptr;
    const unsigned long long *seed_p = seed_dev ? seed_dev : reinterpret_cast<const unsigned long long *>(gamma);
    const int32_t *vl_p = has_vl ? valid_lens : reinterpret_cast<const int32_t *>(gamma);
    const float *bias_p = has_bias ? bias : gamma;
    const int trows = has_vl ? max(Trows, 1) : 1;
    const unsigned long long seed_add = *seed_p;
    float gv[8], bt[8], bv[8], rv[4][8];
    int vl[4];
    ld8(gamma + c, gv);
    ld8(beta + c, bt);
    ld8(bias_p + c, bv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long row = min(m0 + 8 * q + 2 * wave + hf, M - 1);
        ld8(res + row * D + c, rv[q]);
        vl[q] = vl_p[has_vl ? row / trows : 0];
    }

    // ---- all four k-tiles by LDS-DMA: piece p = wave + 4 i covers LDS rows 8 p .. 8 p + 7 of the k-tile (rows 0-31: A, 32-287: W) ----
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int i = 0; i < LL_PCS; ++i) {
            const int pc = i * 4 + wave, byte = pc * 1024 + lane * 16, lrow = byte / LL_ROW, pos = (byte % LL_ROW) / 16;
            const int chunk = (pos ^ ((lrow >> 1) & 7)) * 8 + kt * 64;
            const bf16_t *src = lrow < LL_BM ? A + min(m0 + lrow, M - 1) * lda + chunk : W + (long long)(lrow - LL_BM) * ldw + chunk;
            ll_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt * LL_SLOT + pc * 1024)));
        }
    }
    int f_off[4];       // fragment offsets: row = blk + r (blocks are 32 rows apart: the XOR term is the lane's own), chunk (2 s + hh) ^ ((row >> 1) & 7)
    {
        const int v = (r >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) f_off[s] = r * LL_ROW + (((2 * s + hh) ^ v) << 4);
    }
    f32x16 acc[2];
    acc[0] = (f32x16){0};
    acc[1] = (f32x16){0};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LL_PCS) : "memory");
        else if (kt == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LL_PCS) : "memory");
        else if (kt == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LL_PCS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char *as = smem + kt * LL_SLOT, *bs = as + (LL_BM + wave * 64) * LL_ROW;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(as + f_off[s]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(bs + f_off[s]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(bs + 32 * LL_ROW + f_off[s]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[1], 0, 0, 0);
        }
    }
    __syncthreads();    // every fragment read is done: k-tile 0's space becomes the fp32 tile [32][LL_LDF]
    float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) tile[((g & 3) + 8 * (g >> 2) + 4 * hh) * LL_LDF + wave * 64 + 32 * j + r] = acc[j][g];
    __syncthreads();

    // ---- the row pass: add_layernorm_fwd_kernel<bf16_t, 1, true>, with x taken from the tile and rounded to bf16 first ----------------
    if (seed_dev) seed += seed_add;
    const unsigned thr = drop_thr16(p);
    const DropKey dk = drop_key(seed);
    const float ks = drop_scale16(thr);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rl = 8 * q + 2 * wave + hf;
        long long row = m0 + rl;
        const bool row_valid = row < M;
        if (!row_valid) row = M - 1;
        const float4 t_lo = *reinterpret_cast<const float4 *>(tile + rl * LL_LDF + c), t_hi = *reinterpret_cast<const float4 *>(tile + rl * LL_LDF + c + 4);
        const float xa[8] = {t_lo.x, t_lo.y, t_lo.z, t_lo.w, t_hi.x, t_hi.y, t_hi.z, t_hi.w};
        const bool live = !has_vl || ((int)(row % trows) < vl[q]);
        float v[8], sum = 0.f;
        {
            const unsigned long long idx = (unsigned long long)row * D + c;
            const unsigned km = p > 0.f ? drop_keep_mask<8>((unsigned long long)idx, dk, thr) : ~0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (float)(bf16_t)xa[j] + (has_bias ? bv[j] : 0.f);
                if (p > 0.f) t = ((km >> j) & 1u) ? t * ks : 0.f;
                t = live ? t * alpha : 0.f;
                t += rv[q][j];
                t = (float)(bf16_t)t;   // statistics of the STORED (rounded) row, as a separate LN would see
                v[j] = t;
                sum += t;
            }
            if (row_valid) st8(s_out + row * D + c, v);
        }
        const float mu = half_wave_sum(sum) / D;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; qq += d * d; }
        const float rs = rsqrtf(half_wave_sum(qq) / D + eps);
        if (row_valid) {
            if (l == 0) { mean[row] = mu; rstd[row] = rs; }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * gv[j] + bt[j];
            st8(y + row * D + c, o);
        }
    }
}

}  // namespace

extern "C" {

/* 1 when tsasr_linear_add_layernorm_fwd takes the shape: bf16, N = K = 256, 16-byte aligned rows. */
int tsasr_linear_add_layernorm_ok(long long M, int N, int K, long long lda, long long ldw) {
    return M > 0 && N == LL_D && K == 256 && lda % 8 == 0 && ldw % 8 == 0;
}

/* (s, y, mean, rstd) of tsasr_add_layernorm_fwd applied to x = A[M,K] . W[N,K]^T (bf16, never written), N = K = 256: the GEMM of
 * tsasr_gemm_bf16 (trans_a = 0, trans_b = 0 in its NT reading) and the row kernel in one launch, same bits as the pair.
 * bias (fp32 [N], may be NULL), res / s / y [M,N] bf16 contiguous, valid_lens int32 [M / Trows] (may be NULL). */
int tsasr_linear_add_layernorm_fwd(const void *A, long long lda, const void *W, long long ldw, const float *bias, const void *res, void *s,
                                   void *y, float *mean, float *rstd, const float *gamma, const float *beta, long long M, int N, int K,
                                   float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev,
                                   const int32_t *valid_lens, int Trows, float eps, void *stream) {
    TSASR_CHECK_ARG(A && W && res && s && y && mean && rstd && gamma && beta, "tsasr_linear_add_layernorm_fwd: null pointer");
    TSASR_CHECK_ARG(tsasr_linear_add_layernorm_ok(M, N, K, lda, ldw), "tsasr_linear_add_layernorm_fwd: M=%lld N=%d K=%d lda=%lld ldw=%lld not supported (N = K = 256)",
                    M, N, K, lda, ldw);
    TSASR_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)res | (uintptr_t)s | (uintptr_t)y) & 15) == 0, "tsasr_linear_add_layernorm_fwd: misaligned pointer");
    TSASR_CHECK_ARG(p >= 0.f && p < 1.f, "tsasr_linear_add_layernorm_fwd: bad p");
    TSASR_CHECK_ARG(!valid_lens || (Trows > 0 && M % Trows == 0), "tsasr_linear_add_layernorm_fwd: rows not a multiple of T");
    const int lds = 4 * LL_SLOT;
    (void)hipFuncSetAttribute((const void *)linear_add_layernorm_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    linear_add_layernorm_fwd_kernel<<<(unsigned)((M + LL_BM - 1) / LL_BM), 256, lds, (hipStream_t)stream>>>(
        (const bf16_t *)A, lda, (const bf16_t *)W, ldw, bias, (const bf16_t *)res, (bf16_t *)s, (bf16_t *)y, mean, rstd, gamma, beta, M, alpha, p, seed,
        seed_dev, valid_lens, Trows, eps);
    TSASR_CHECK_LAUNCH("tsasr_linear_add_layernorm_fwd");
    return 0;
}

}  // extern "C"
