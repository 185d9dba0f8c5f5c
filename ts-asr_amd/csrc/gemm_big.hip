// Wide-output bf16 GEMM for gfx950: C[M,N] = epilogue(A[M,K] . B[N,K]^T) with 256-column output tiles and 8 waves per workgroup.
//
// Replaces, for the macaron feed-forward's up-projection and for the data gradient behind its down-projection
// (SB/nnet/attention.py:820-836 `PositionalwiseFeedForward.ffn`: Linear(D, 4D... d_ffn) -> activation -> Dropout -> Linear(d_ffn, D);
// M = B*T' = 8000 rows, N = d_ffn = 2048, K = d_model = 256), the 128x128 / 4-wave tiles of csrc/gemm.hip. At K = 256 such a GEMM
// is not MFMA-bound but bound by what a CU can take in per second (LDS-DMA from L2 / Infinity Cache: ~40 GB/s per CU with every CU
// loading, profiles/r02_notes.md): 1008 tiles of 128x128 move 129 MB through the CUs' load paths, 256 tiles of 256x256 move 67 MB.
//   workgroup = 512 threads = 8 waves as 2 (M) x 4 (N); wave tile (BM/2) x 64 of v_mfma_f32_32x32x16_bf16 blocks
//   k-tile = [BM + 256 rows][64 k] (128-byte rows) filled by LDS-DMA issued as inline asm (see csrc/wgrad.hip: through the builtin
//   hipcc drains the DMA in front of the fragment reads), XOR swizzle on the SOURCE chunk, two slots, counted waits, raw s_barrier
//   epilogue modes (same contract as csrc/gemm.hip's EpiArgs):
//     0: C = acc                                 1: C = dropout_p(LeakyReLU(acc + bias[n]))
//     2: C = acc * keep(m,n)/(1-p) * LeakyReLU'(y[m,n]); per-tile column sums -> colpart (the bias gradient)
//   the accumulators go through LDS as fp32 half tiles and leave row-major, 8 columns per thread (see the epilogue).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define GBG_BN 256
#define GBG_BK 64
#define GBG_THREADS 512
#define GBG_ROW 128          // bytes per k-contiguous LDS row (64 bf16)
#define GBG_LDF (GBG_BN + 4) // epilogue tile row stride in floats (1040 B)

struct BigArgs {
    const bf16_t *A, *B;
    bf16_t *C;
    int M, N, K;
    long long lda, ldb, ldc;
    int mode;
    const float *bias;
    const bf16_t *y;
    long long ldy;
    float slope, p;
    unsigned long long seed;
    const unsigned long long *seed_dev;
    float *colpart;   // [tiles_m][N]
    // one 16-bit word per 8 consecutive outputs, [M][N/8]: bits 0-7 = dropout keep-bits, bits 8-15 = "stored activation is negative".
    // Mode 1 writes it (when given); mode 2 reads it INSTEAD of the saved activation y and of re-hashing the keep-bits: 4 MB instead of
    // 32.8 MB read per FFN backward launch at configs[1], no hash (tools/gemm_bench.py: the mode-2 epilogue was 13 us of a 33 us launch)
    unsigned short *mask;
};

__device__ __forceinline__ void gbg_dma16(const bf16_t *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int BM, int MODE, bool MASK = false>   // MASK: the epilogue's mask words are written (mode 1) / read instead of y and the hash (mode 2)
__global__ __launch_bounds__(GBG_THREADS, 2) void gemm_big_kernel(BigArgs a) {
    constexpr int A_BYTES = BM * GBG_ROW, B_BYTES = GBG_BN * GBG_ROW, SLOT = A_BYTES + B_BYTES;
    constexpr int A_PCS = A_BYTES / 1024 / 8, B_PCS = B_BYTES / 1024 / 8;   // 1 KiB DMA pieces per wave per k-tile
    constexpr int RB = BM / 64;                                             // 32-row blocks per wave (wave tile = (BM/2) x 64)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, hh = lane >> 5;
    const int tiles_n = a.N / GBG_BN, tiles_m = (a.M + BM - 1) / BM, nwg = tiles_n * tiles_m;
    int id;
    {   // XCD-contiguous tile ids (guide T1): the tiles of one row panel share an XCD's L2
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int tn = id % tiles_n, tm = id / tiles_n, m0 = tm * BM, n0 = tn * GBG_BN;
    const int nk = a.K / GBG_BK;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // DMA sources of k-tile 0: piece p covers LDS rows 8p .. 8p+7 (128 B each); LDS slot (row, pos) takes global chunk pos ^ ((row >> 1) & 7)
    const bf16_t *ga[A_PCS], *gb[B_PCS];
#pragma unroll
    for (int i = 0; i < A_PCS; ++i) {
        const int byte = (i * 8 + wave) * 1024 + lane * 16, lrow = byte / GBG_ROW, pos = (byte % GBG_ROW) / 16;
        ga[i] = a.A + (long long)min(m0 + lrow, a.M - 1) * a.lda + (pos ^ ((lrow >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PCS; ++i) {
        const int byte = (i * 8 + wave) * 1024 + lane * 16, lrow = byte / GBG_ROW, pos = (byte % GBG_ROW) / 16;
        gb[i] = a.B + (long long)(n0 + lrow) * a.ldb + (pos ^ ((lrow >> 1) & 7)) * 8;
    }
    auto issue = [&](int kt, int slot) {
        const unsigned base = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < A_PCS; ++i) gbg_dma16(ga[i] + kt * GBG_BK, __builtin_amdgcn_readfirstlane(base + (unsigned)(i * 8 + wave) * 1024u));
#pragma unroll
        for (int i = 0; i < B_PCS; ++i) gbg_dma16(gb[i] + kt * GBG_BK, __builtin_amdgcn_readfirstlane(base + A_BYTES + (unsigned)(i * 8 + wave) * 1024u));
    };
    // fragment offsets: row = blk + r, chunk (2s + hh) ^ ((row >> 1) & 7); blocks are 32 rows apart, so the XOR term is the lane's own
    int f_off[GBG_BK / 16];
    {
        const int v = (r >> 1) & 7;
#pragma unroll
        for (int s = 0; s < GBG_BK / 16; ++s) f_off[s] = r * GBG_ROW + (((2 * s + hh) ^ v) << 4);
    }
    f32x16 acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char *as = smem + (kt & 1) * SLOT + wm * (BM / 2) * GBG_ROW, *bs = smem + (kt & 1) * SLOT + A_BYTES + wn * 64 * GBG_ROW;
#pragma unroll
        for (int s = 0; s < GBG_BK / 16; ++s) {
            bf16x8 af[RB], bfr[2];
#pragma unroll
            for (int i = 0; i < RB; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * GBG_ROW + f_off[s]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(bs + j * 32 * GBG_ROW + f_off[s]);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // every fragment read is done: the ring becomes the fp32 epilogue tile [BM/2][GBG_LDF]
    // Epilogue in two halves of BM/2 rows: the waves that own the half put their fp32 accumulators into LDS (lane = column: conflict-free
    // 4-byte stores), then ALL threads walk the half row-major, 8 consecutive columns per thread: one hash per column pair, the bias /
    // saved activation as 16-byte vectors, one 16-byte store. (Applied in the accumulator layout - lane = column, 128 rows per lane - the
    // same arithmetic cost 10 us (mode 1) and 19 us (mode 2) on top of a 18 us GEMM: one hash per element, 2-byte LDS accesses.)
    float *tile = reinterpret_cast<float *>(smem);
    unsigned long long seed = a.seed;
    if (MODE != 0 && a.seed_dev) seed += *a.seed_dev;
    const unsigned thr = drop_thr16(a.p);
    const float ks = drop_scale16(thr);
    const float ks_eff = a.p > 0.f ? ks : 1.f, slope_eff = a.slope >= 0.f ? a.slope : 1.f;     // (the launcher refuses slopes above 1)
    const unsigned ks_bits = __float_as_uint(ks_eff), ks_slope_bits = __float_as_uint(ks_eff * slope_eff);
    const DropKey dk = drop_key(seed);
    constexpr int HR = BM / 2, CPR = GBG_BN / 8, NIT = HR * CPR / GBG_THREADS;   // rows per half, 16-byte chunks per row, chunks per thread
    const int cc = (threadIdx.x % CPR) * 8, rr0 = threadIdx.x / CPR, n = n0 + cc;   // GBG_THREADS % CPR == 0: one column group per thread
    float bias8[8], csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
        if (a.bias) ld8(a.bias + n, bias8);     // one guarded block, two 16-byte loads (eight guarded scalar loads were eight round trips)
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] *= ks_eff;     // the dropout scale folded into the bias (see the epilogue)
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        constexpr bool use_mask = MODE == 2 && MASK;
        uint4 yraw[use_mask ? 1 : NIT];
        unsigned short mraw[use_mask ? NIT : 1];
        if (MODE == 2) {   // requested before the accumulators go through LDS
            if constexpr (use_mask) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    mraw[it] = a.mask[((long long)min(m0 + half * HR + rr0 + it * (GBG_THREADS / CPR), a.M - 1) * a.N + n) >> 3];
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    yraw[it] = *reinterpret_cast<const uint4 *>(a.y + (long long)min(m0 + half * HR + rr0 + it * (GBG_THREADS / CPR), a.M - 1) * a.ldy + n);
            }
        }
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        tile[(32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * GBG_LDF + wn * 64 + 32 * j + r] = acc[i][j][g];
        }
        __syncthreads();
        // FULL (every row of the half tile inside the matrix: all tiles but the last row panel) stores without per-row guards: behind
        // an exec-masked guard each of the 8 stores of a pass was waited for before the next pass started (s_waitcnt vmcnt(0) at the
        // head of every guarded block): 16 serialized store round trips per workgroup
        auto pass = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rr0 + it * (GBG_THREADS / CPR), m = m0 + half * HR + rr;
            const float4 v_lo = *reinterpret_cast<const float4 *>(tile + rr * GBG_LDF + cc), v_hi = *reinterpret_cast<const float4 *>(tile + rr * GBG_LDF + cc + 4);
            float v[8] = {v_lo.x, v_lo.y, v_lo.z, v_lo.w, v_hi.x, v_hi.y, v_hi.z, v_hi.w};
            if (MODE != 0) {
                unsigned km;
                if constexpr (use_mask) km = (unsigned)mraw[it];
                else km = a.p > 0.f ? drop_keep_mask<8>((unsigned long long)m * a.N + n, dk, thr) : ~0u;
                if (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // dropout(lrelu(x)) = lrelu(x * ks) & keep for ks > 0 (positive homogeneity): the scale rides in the bias add
                        // (one fma), LeakyReLU with 0 <= slope <= 1 is max(x, slope * x) (no activation: slope 1), a dropped element
                        // is +0 (keep-bit spread over a word by v_bfe_i32, ANDed in): 5 operations per element instead of 8 - each one
                        // is 0.85 us of this launch at 8000 x 2048 (256 outputs per thread, two waves per SIMD)
                        const float x = __builtin_fmaf(v[e], ks_eff, bias8[e]);
                        const float t = fmaxf(x, x * slope_eff);
                        v[e] = __uint_as_float(__float_as_uint(t) & (unsigned)((int)(km << (31 - e)) >> 31));
                    }
                    if constexpr (MASK) {     // sign bits of the values AS STORED (bf16, pair-converted; -0 counts as not negative)
                        unsigned neg = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned w = pk_bf16(v[2 * q], v[2 * q + 1]);
                            neg |= ((w & 0xffffu) > 0x8000u ? 1u : 0u) << (2 * q) | (w > 0x80000000u ? 1u : 0u) << (2 * q + 1);
                        }
                        if (FULL || m < a.M) a.mask[((long long)m * a.N + n) >> 3] = (unsigned short)((km & 0xffu) | (neg << 8));
                    }
                } else {
                    // mode 2: keep-bits (0 .. 7) and "the stored activation was negative" bits (8 .. 15) of the 8 elements - the forward's mask
                    // word, or the same word rebuilt from the hash and the saved activation; ONE arithmetic path for both (they must give
                    // the same bits). Per element: the factor ks or ks * slope picked by the sign bit (v_bfe_i32 + v_bfi_b32), one
                    // multiply, the keep-bit ANDed in (a dropped element is +0).
                    unsigned kw = km & 0xffu;
                    if constexpr (use_mask) kw = km;
                    else {
                        const unsigned yw[4] = {yraw[it].x, yraw[it].y, yraw[it].z, yraw[it].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)       // sign bit of a bf16 activation that is not a zero
                            kw |= (((yw[q] & 0xffffu) > 0x8000u ? 1u : 0u) << (8 + 2 * q)) | ((yw[q] > 0x80000000u ? 1u : 0u) << (9 + 2 * q));
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned ng = (unsigned)((int)(kw << (23 - e)) >> 31);            // the activation was negative: all ones
                        float t = v[e] * __uint_as_float((ng & ks_slope_bits) | (~ng & ks_bits));
                        t = __uint_as_float(__float_as_uint(t) & (unsigned)((int)(kw << (31 - e)) >> 31));
                        asm("" : "+v"(t));      // the ROUNDED product joins the column sum in every instantiation (no contraction into an fma here and not there)
                        v[e] = t;
                        if (FULL || m < a.M) csum[e] += t;
                    }
                }
            }
            if (FULL || m < a.M) st8_g(a.C + (long long)m * a.ldc + n, v);    // write-through: 32.8 MB per launch that the next kernel reads from beyond L2
        }
        };
        if (m0 + half * HR + HR <= a.M) pass(std::true_type{});
        else pass(std::false_type{});
        __syncthreads();
    }
    if (MODE == 2 && a.colpart) {   // column sums of this tile: per-thread partials -> LDS -> one row of colpart (fixed order)
        float *red = tile;           // [GBG_THREADS / CPR][GBG_LDF]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rr0 * GBG_LDF + cc + e] = csum[e];
        __syncthreads();
        if (threadIdx.x < GBG_BN) {
            float sum = 0.f;
#pragma unroll 4
            for (int q = 0; q < GBG_THREADS / CPR; ++q) sum += red[q * GBG_LDF + threadIdx.x];
            a.colpart[(long long)tm * a.N + n0 + threadIdx.x] = sum;
        }
    }
}

// rows per tile for an M x N problem: 256 when that still gives >= 200 workgroups, else 128; 0 = this kernel does not apply
int tsasr_gemm_big_bm(int M, int N, int K) {
    static const int on = getenv("TSASR_GEMM_BIG") ? atoi(getenv("TSASR_GEMM_BIG")) : 1;
    if (!on || N % GBG_BN != 0 || K % GBG_BK != 0 || N < 1024 || K > 1024 || M < 256) return 0;
    const int tn = N / GBG_BN;
    if (cdiv(M, 256) * tn >= 200) return 256;
    if (cdiv(M, 128) * tn >= 96) return 128;
    return 0;
}

size_t tsasr_gemm_big_colpart_rows(int M, int N, int K) {
    const int bm = tsasr_gemm_big_bm(M, N, K);
    return bm ? (size_t)cdiv(M, bm) : 0;
}

int tsasr_gemm_big_launch(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int mode,
                          const float *bias, const void *y, long long ldy, float slope, float p, unsigned long long seed,
                          const unsigned long long *seed_dev, float *colpart, void *mask, hipStream_t st) {
    const int bm = tsasr_gemm_big_bm(M, N, K);
    if (!bm) return 1;
    if (mode == 1 && slope > 1.f) return 1;      // the mode-1 epilogue's LeakyReLU is max(x, slope * x): the caller's general path takes slopes above 1
    BigArgs a{(const bf16_t *)A, (const bf16_t *)B, (bf16_t *)C, M, N, K, lda, ldb, ldc, mode, bias, (const bf16_t *)y, ldy, slope, p, seed, seed_dev, colpart,
              (unsigned short *)mask};
    const int grid = cdiv(M, bm) * (N / GBG_BN);
    constexpr int LDS256 = 128 * GBG_LDF * 4 > 2 * (256 + GBG_BN) * GBG_ROW ? 128 * GBG_LDF * 4 : 2 * (256 + GBG_BN) * GBG_ROW;   // fp32 half tile 130 KiB
    constexpr int LDS128 = 2 * (128 + GBG_BN) * GBG_ROW;   // 96 KiB >= the 64 x 260 fp32 half tile
    void (*kern)(BigArgs) = nullptr;
    if (mask && mode != 0) {
        if (bm == 256) kern = mode == 1 ? gemm_big_kernel<256, 1, true> : gemm_big_kernel<256, 2, true>;
        else kern = mode == 1 ? gemm_big_kernel<128, 1, true> : gemm_big_kernel<128, 2, true>;
    } else if (bm == 256) kern = mode == 0 ? gemm_big_kernel<256, 0> : mode == 1 ? gemm_big_kernel<256, 1> : gemm_big_kernel<256, 2>;
    else kern = mode == 0 ? gemm_big_kernel<128, 0> : mode == 1 ? gemm_big_kernel<128, 1> : gemm_big_kernel<128, 2>;
    const int lds = bm == 256 ? LDS256 : LDS128;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    kern<<<grid, GBG_THREADS, lds, st>>>(a);
    return 0;
}
