// bf16 MFMA GEMM for the Conformer block's projections on gfx950, with the three operand layouts a Linear layer needs and a
// "accumulate into the fp32 gradient arena" epilogue.
//
// Replaces the hipBLASLt calls behind torch.nn.functional.linear for (SB = vendor/speechbrain/speechbrain)
//   PositionalwiseFeedForward SB/nnet/attention.py:820-836, RelPosMHAXL in/out/pos projections attention.py:549-553,581-583,635,
//   ConvolutionModule pointwise convs Conformer.py:76-82,98, Linear SB/nnet/linear.py:64-78 (encoder/decoder/speaker projections).
// At the model's shapes (M = B*T' = 8000 rows, N,K in {256, 512, 768, 2048, 2560}) the library picks 64x64 macro-tiles that
// reach ~150 TFLOP/s (profiles/r01_*); and every weight gradient costs two more elementwise launches (bf16 -> fp32 cast, add
// into .grad). Here:
//   C[M,N] (+)= op(A)[M,K] . op(B)[K,N]      A: [M,K] (transA=0) or [K,M] (transA=1); B: [N,K] (transB=0) or [K,N] (transB=1)
//   fwd   y  = x . W^T      : transA=0, transB=0        (both operands k-contiguous: ds_read_b128 fragments)
//   dgrad dx = dy . W       : transA=0, transB=1        (B fragments through ds_read_b64_tr_b16)
//   wgrad dW += dy^T . x    : transA=1, transB=1, fp32 output ACCUMULATED into the gradient arena, split along the long
//                             inner dimension (M) into per-workgroup fp32 slabs that a second kernel adds in a fixed order
//                             (deterministic; no float atomics).
// Kernel: 256 threads = 2x2 waves, macro-tile BMxBN in {128x128, 64x64} (picked so that the grid covers the 256 CUs), BK = 64,
// v_mfma_f32_32x32x16_bf16, fp32 accumulators, two LDS buffers with the next tile's global loads in flight during the MFMAs
// (register staging, one barrier per k-tile), LDS rows padded by 16 B (conflict-free 16-byte fragment reads).
#include "common.h"

#define GB_K 64

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

template <int BM, int BN, bool AT, bool BT>
struct GemmSmem {
    // non-transposed operand tile: [rows][GB_K + 8]; transposed: [GB_K][rows + 8]
    static constexpr int A_LD = AT ? (BM + 8) : (GB_K + 8);
    static constexpr int B_LD = BT ? (BN + 8) : (GB_K + 8);
    static constexpr int A_ELEMS = AT ? GB_K * A_LD : BM * A_LD;
    static constexpr int B_ELEMS = BT ? GB_K * B_LD : BN * B_LD;
    static constexpr int STAGE = A_ELEMS + B_ELEMS;
    static constexpr size_t BYTES = (size_t)2 * STAGE * sizeof(bf16_t);
};

// one 16-byte chunk (8 bf16) of an operand tile per (thread, iteration): global -> registers
template <int ROWS, bool TR>
struct TileLoader {
    // TR=false: tile is [ROWS][GB_K] of a [rows, K] matrix (ld = leading dim); TR=true: tile is [GB_K][ROWS] of a [K, rows] matrix
    static constexpr int CHUNKS = ROWS * GB_K / 8;
    static constexpr int ITERS = (CHUNKS + 255) / 256;
    uint4 v[ITERS];
    __device__ __forceinline__ void load(const bf16_t *__restrict__ src, long long ld, int row0, int nrows, int k0, int K) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = it * 256 + threadIdx.x;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (c < CHUNKS) {
                if (!TR) {
                    const int rr = c / (GB_K / 8), kk = (c % (GB_K / 8)) * 8;
                    if (row0 + rr < nrows && k0 + kk < K) x = *reinterpret_cast<const uint4 *>(src + (long long)(row0 + rr) * ld + k0 + kk);
                } else {
                    const int kk = c / (ROWS / 8), rr = (c % (ROWS / 8)) * 8;
                    if (k0 + kk < K && row0 + rr < nrows) x = *reinterpret_cast<const uint4 *>(src + (long long)(k0 + kk) * ld + row0 + rr);
                }
            }
            v[it] = x;
        }
    }
    __device__ __forceinline__ void store(bf16_t *lds, int LD) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = it * 256 + threadIdx.x;
            if (c < CHUNKS) {
                if (!TR) {
                    const int rr = c / (GB_K / 8), kk = (c % (GB_K / 8)) * 8;
                    *reinterpret_cast<uint4 *>(lds + rr * LD + kk) = v[it];
                } else {
                    const int kk = c / (ROWS / 8), rr = (c % (ROWS / 8)) * 8;
                    *reinterpret_cast<uint4 *>(lds + kk * LD + rr) = v[it];
                }
            }
        }
    }
};

// MFMA operand fragment for k-step s (16 deep) of the 32-wide block starting at `blk0` of the tile
template <bool TR>
__device__ __forceinline__ bf16x8 frag(const bf16_t *lds, int LD, int blk0, int s, int lane) {
    const int r = lane & 31, hh = lane >> 5;
    if (!TR) return *reinterpret_cast<const bf16x8 *>(lds + (blk0 + r) * LD + 16 * s + 8 * hh);
    const int mhalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const bf16_t *a0 = lds + (16 * s + 8 * hh + q4) * LD + blk0 + 16 * mhalf + 4 * p4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a0 + 4 * LD));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

// OUT_MODE: 0 = bf16 store, 1 = fp32 store (slab or plain), 2 = fp32 accumulate (C += acc)
template <int BM, int BN, bool AT, bool BT, int OUT_MODE>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, void *__restrict__ Cv,
                                                        int M, int N, int K, long long lda, long long ldb, long long ldc, int kchunk,
                                                        long long slab_stride) {
    using S = GemmSmem<BM, BN, AT, BT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t *lds = reinterpret_cast<bf16_t *>(smem);
    constexpr int RB = BM / 64, CB = BN / 64;  // 32x32 blocks per wave in each direction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = (f32x16){0};

    TileLoader<BM, AT> la;
    TileLoader<BN, BT> lb;
    const int nk = (kend - kbeg + GB_K - 1) / GB_K;
    if (nk > 0) {
        la.load(A, lda, m0, M, kbeg, kend);
        lb.load(B, ldb, n0, N, kbeg, kend);
        la.store(lds, S::A_LD);
        lb.store(lds + S::A_ELEMS, S::B_LD);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        bf16_t *cur = lds + (kt & 1) * S::STAGE, *nxt = lds + ((kt + 1) & 1) * S::STAGE;
        const bool more = kt + 1 < nk;
        if (more) {
            la.load(A, lda, m0, M, kbeg + (kt + 1) * GB_K, kend);
            lb.load(B, ldb, n0, N, kbeg + (kt + 1) * GB_K, kend);
        }
        const bf16_t *as = cur, *bs = cur + S::A_ELEMS;
#pragma unroll
        for (int s = 0; s < GB_K / 16; ++s) {
            bf16x8 af[RB], bfr[CB];
#pragma unroll
            for (int i = 0; i < RB; ++i) af[i] = frag<AT>(as, S::A_LD, wm * (BM / 2) + 32 * i, s, lane);
#pragma unroll
            for (int j = 0; j < CB; ++j) bfr[j] = frag<BT>(bs, S::B_LD, wn * (BN / 2) + 32 * j, s, lane);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            la.store(nxt, S::A_LD);
            lb.store(nxt + S::A_ELEMS, S::B_LD);
        }
        __syncthreads();
    }
    // epilogue: acc[i][j][g] -> C[m0 + wm*BM/2 + 32i + row(g)][n0 + wn*BN/2 + 32j + (lane&31)]
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int n = n0 + wn * (BN / 2) + 32 * j + r;
            if (n >= N) continue;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int m = m0 + wm * (BM / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh;
                if (m >= M) continue;
                if (OUT_MODE == 0) reinterpret_cast<bf16_t *>(Cv)[(long long)m * ldc + n] = (bf16_t)acc[i][j][g];
                else if (OUT_MODE == 1) reinterpret_cast<float *>(Cv)[(long long)blockIdx.z * slab_stride + (long long)m * ldc + n] = acc[i][j][g];
                else reinterpret_cast<float *>(Cv)[(long long)m * ldc + n] += acc[i][j][g];
            }
        }
}

// C[m][n] (+)= sum_z slab[z][m*N + n]   (fixed order)
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ C, int M, int N,
                                                               long long ldc, int nslab, long long slab_stride, int accumulate) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long long)M * N) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nslab; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(slab + (long long)z * slab_stride + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int m = (int)(i / N), n = (int)(i % N);
    float *c = C + (long long)m * ldc + n;
    if (accumulate) { c[0] += s.x; c[1] += s.y; c[2] += s.z; c[3] += s.w; }
    else { c[0] = s.x; c[1] = s.y; c[2] = s.z; c[3] = s.w; }
}

template <int BM, int BN, bool AT, bool BT, int OUT_MODE>
static void launch(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int splits,
                   int kchunk, long long slab_stride, hipStream_t st) {
    using S = GemmSmem<BM, BN, AT, BT>;
    auto kern = gemm_bf16_kernel<BM, BN, AT, BT, OUT_MODE>;
    if (S::BYTES > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), splits);
    kern<<<grid, 256, S::BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, C, M, N, K, lda, ldb, ldc, kchunk, slab_stride);
}

template <int BM, int BN, int OUT_MODE>
static void launch_t(int tA, int tB, const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                     int splits, int kchunk, long long slab_stride, hipStream_t st) {
    if (!tA && !tB) launch<BM, BN, false, false, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st);
    else if (!tA && tB) launch<BM, BN, false, true, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st);
    else if (tA && tB) launch<BM, BN, true, true, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st);
    else launch<BM, BN, true, false, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st);
}

struct GemmPlan { int big; int splits; int kchunk; };
static GemmPlan plan(int M, int N, int K, int out_f32) {
    GemmPlan p;
    const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128), t64 = (long long)cdiv(M, 64) * cdiv(N, 64);
    p.big = t128 >= 192;                       // enough 128x128 tiles to cover the chip; otherwise 64x64 tiles
    const long long tiles = p.big ? t128 : t64;
    p.splits = 1;
    if (out_f32 && tiles < 256 && K >= 1024) {  // weight gradients: few output tiles, long inner dimension -> split it
        int s = (int)((512 + tiles - 1) / tiles);
        const int max_s = K / 256;             // at least 4 k-tiles per split
        if (s > max_s) s = max_s;
        if (s > 32) s = 32;
        if (s < 1) s = 1;
        p.splits = s;
    }
    p.kchunk = cdiv(cdiv(K, p.splits), GB_K) * GB_K;
    p.splits = cdiv(K, p.kchunk);
    return p;
}

extern "C" {

size_t tsasr_gemm_bf16_workspace_bytes(int M, int N, int K, int out_dtype) {
    const GemmPlan p = plan(M, N, K, out_dtype == TSASR_F32);
    return p.splits > 1 ? align_up((size_t)p.splits * M * N * sizeof(float), 256) : 0;
}

/* C[M,N] (+)= op(A)[M,K] . op(B)[K,N], bf16 operands, fp32 accumulation.
 *   transA = 0: A is [M,K] row-major (lda);  transA = 1: A is [K,M] row-major
 *   transB = 0: B is [N,K] row-major (ldb) - the layout of a Linear weight for y = x.W^T;  transB = 1: B is [K,N] row-major
 *   out_dtype TSASR_BF16 (store) or TSASR_F32 (store, or accumulate != 0: C += result - gradient-arena epilogue).
 * Leading dimensions and K must be multiples of 8 elements (16-byte rows); pointers 16-byte aligned. */
int tsasr_gemm_bf16(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                    int transA, int transB, int out_dtype, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(A && B && C, "tsasr_gemm_bf16: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && K > 0, "tsasr_gemm_bf16: empty problem (%d,%d,%d)", M, N, K);
    TSASR_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && (transA ? M : K) % 8 == 0 && (transB ? N : K) % 8 == 0,
                    "tsasr_gemm_bf16: rows must be multiples of 8 bf16 (M=%d N=%d K=%d lda=%lld ldb=%lld)", M, N, K, lda, ldb);
    TSASR_CHECK_ARG(out_dtype == TSASR_BF16 || out_dtype == TSASR_F32, "tsasr_gemm_bf16: bad out_dtype %d", out_dtype);
    TSASR_CHECK_ARG(!(accumulate && out_dtype != TSASR_F32), "tsasr_gemm_bf16: accumulate needs fp32 output");
    const GemmPlan p = plan(M, N, K, out_dtype == TSASR_F32);
    hipStream_t st = (hipStream_t)stream;
    if (p.splits > 1) {
        TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_gemm_bf16_workspace_bytes(M, N, K, out_dtype), "tsasr_gemm_bf16: workspace too small");
        TSASR_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "tsasr_gemm_bf16: split-K output needs N, ldc multiples of 4");
        const long long ss = (long long)M * N;
        if (p.big) launch_t<128, 128, 1>(transA, transB, A, B, workspace, M, N, K, lda, ldb, N, p.splits, p.kchunk, ss, st);
        else launch_t<64, 64, 1>(transA, transB, A, B, workspace, M, N, K, lda, ldb, N, p.splits, p.kchunk, ss, st);
        gemm_slab_reduce_kernel<<<(unsigned)cdiv((int)((ss + 3) / 4), 256), 256, 0, st>>>((const float *)workspace, (float *)C, M, N, ldc, p.splits, ss, accumulate);
    } else if (out_dtype == TSASR_BF16) {
        if (p.big) launch_t<128, 128, 0>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
        else launch_t<64, 64, 0>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    } else if (accumulate) {
        if (p.big) launch_t<128, 128, 2>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
        else launch_t<64, 64, 2>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    } else {
        if (p.big) launch_t<128, 128, 1>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
        else launch_t<64, 64, 1>(transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    }
    TSASR_CHECK_LAUNCH("tsasr_gemm_bf16");
    return 0;
}

}  // extern "C"
