// fp32 GEMM for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): the parity mode of the TS-ASR path.
//
// The reference's default precision is fp32 (hparams/LibriSpeechMix/conformer-t_scratch.yaml:88): the Linear layers behind
// speechbrain/nnet/linear.py:64-78, attention.py:549-553, 581-583, 820-836 and Conformer.py:76-82, 98 are fp32 GEMMs there. The
// benchmarked step computes them in bf16 (csrc/gemm.hip); this kernel is what `--dtype fp32` / the fp32 stage tests run them on -
// products and sums in fp32, no rounding of the operands - so that the fp32 mode exercises hand-written code as well instead of the
// library GEMM. Same operand conventions as tsasr_gemm_bf16:
//   C[M,N] (+)= op(A)[M,K] . op(B)[K,N];  transA=0: A [M,K], 1: A [K,M];  transB=0: B [N,K] (a Linear weight), 1: B [K,N].
// 64x64 output tile, 4 waves (2x2) of one 32x32 accumulator each, 16-wide k-tiles through LDS (padded rows); every global operand is
// requested unconditionally (row / column clamped into the matrix) and masked afterwards. Built for exactness and every shape, not
// for speed: ~40 TFLOP/s of the 157 the fp32 matrix cores have.
#include "common.h"

#define GF_T 64
#define GF_K 16
#define GF_LD (GF_K + 1)

template <bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                       int N, int K, long long lda, long long ldb, long long ldc, int accumulate) {
    __shared__ float a_lds[GF_T * GF_LD], b_lds[GF_T * GF_LD];      // [row m / column n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GF_T, n0 = blockIdx.x * GF_T;
    f32x16 acc = {0};
    // element (row, k) of this thread in pass p of a tile load: k-contiguous storage walks k fastest (coalesced 64-byte runs),
    // row-contiguous storage (the transposed forms) walks the row fastest
    auto a_rk = [&](int p, int &row, int &k) {
        if (AT) { row = tid & 63; k = (tid >> 6) + 4 * p; }
        else { k = tid & 15; row = (tid >> 4) + 16 * p; }
    };
    auto b_rk = [&](int p, int &row, int &k) {
        if (BT) { row = tid & 63; k = (tid >> 6) + 4 * p; }
        else { k = tid & 15; row = (tid >> 4) + 16 * p; }
    };
    for (int k0 = 0; k0 < K; k0 += GF_K) {
        float av[4], bv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int row, k;
            a_rk(p, row, k);
            const int m = min(m0 + row, M - 1), kk = min(k0 + k, K - 1);
            av[p] = AT ? A[(long long)kk * lda + m] : A[(long long)m * lda + kk];
            b_rk(p, row, k);
            const int n = min(n0 + row, N - 1), kb = min(k0 + k, K - 1);
            bv[p] = BT ? B[(long long)kb * ldb + n] : B[(long long)n * ldb + kb];
        }
        __syncthreads();                     // the previous tile's fragment reads are done
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int row, k;
            a_rk(p, row, k);
            a_lds[row * GF_LD + k] = (m0 + row < M && k0 + k < K) ? av[p] : 0.f;
            b_rk(p, row, k);
            b_lds[row * GF_LD + k] = (n0 + row < N && k0 + k < K) ? bv[p] : 0.f;
        }
        __syncthreads();
        // v_mfma_f32_32x32x2_f32: lane l supplies A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32]
#pragma unroll
        for (int s = 0; s < GF_K / 2; ++s) {
            const float a = a_lds[(32 * wm + (lane & 31)) * GF_LD + 2 * s + (lane >> 5)];
            const float b = b_lds[(32 * wn + (lane & 31)) * GF_LD + 2 * s + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // accumulator: register g = row (g & 3) + 8 (g >> 2) + 4 (lane >> 5), column lane & 31 of the wave's 32x32 block
    const int n = n0 + 32 * wn + (lane & 31);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int m = m0 + 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) {
            float *c = C + (long long)m * ldc + n;
            *c = accumulate ? *c + acc[g] : acc[g];
        }
    }
}

extern "C" {

/* C[M,N] (+)= op(A) . op(B), all fp32; layouts as tsasr_gemm_bf16 (transA=0: A [M,K], 1: A [K,M]; transB=0: B [N,K], 1: B [K,N]).
 * accumulate != 0: C += result. Any M, N, K >= 1, any strides. */
int tsasr_gemm_f32(const float *A, const float *B, float *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int transA,
                   int transB, int accumulate, void *stream) {
    TSASR_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "tsasr_gemm_f32: bad arguments (M=%d N=%d K=%d)", M, N, K);
    TSASR_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= N, "tsasr_gemm_f32: leading dimension shorter than a row");
    const dim3 grid(cdiv(N, GF_T), cdiv(M, GF_T));
    TSASR_CHECK_ARG(grid.y <= 65535u, "tsasr_gemm_f32: M=%d too large for one launch", M);
    hipStream_t st = (hipStream_t)stream;
    if (!transA && !transB) gemm_f32_kernel<false, false><<<grid, 256, 0, st>>>(A, B, C, M, N, K, lda, ldb, ldc, accumulate);
    else if (!transA && transB) gemm_f32_kernel<false, true><<<grid, 256, 0, st>>>(A, B, C, M, N, K, lda, ldb, ldc, accumulate);
    else if (transA && transB) gemm_f32_kernel<true, true><<<grid, 256, 0, st>>>(A, B, C, M, N, K, lda, ldb, ldc, accumulate);
    else gemm_f32_kernel<true, false><<<grid, 256, 0, st>>>(A, B, C, M, N, K, lda, ldb, ldc, accumulate);
    TSASR_CHECK_LAUNCH("tsasr_gemm_f32");
    return 0;
}

}  // extern "C"
