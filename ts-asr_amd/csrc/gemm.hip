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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "common.h"

#define GB_K 64

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

// Fused epilogues of the bf16-output GEMM (mode 0 = plain store):
//   1: C = dropout_p(LeakyReLU(acc + bias[n]))                       - Linear + activation + Dropout of the macaron FFN
//   2: C = acc * keep(m,n)/(1-p) * LeakyReLU'(y[m,n]) ; colsum -> dbias - the same chain's backward applied to the dgrad GEMM
// The dropout mask is the counter-based one of common.h (index m*N + n, seed + *seed_dev), as in elementwise.hip.
struct EpiArgs {
    int mode;
    const float *bias;
    const bf16_t *y;
    long long ldy;
    float slope, p;
    unsigned long long seed;
    const unsigned long long *seed_dev;
    float *colpart;   // [gridDim.y][N] partial column sums (mode 2)
    // batched form (tsasr_gemm_bf16_nt_batched): workgroups of grid row blockIdx.y multiply A by B = btab[blockIdx.y] and write the C
    // matrix c_batch elements after the previous one; NULL: one product
    const void *const *btab;
    long long c_batch;
};

template <int BM, int BN, bool AT, bool BT>
struct GemmSmem {
    // non-transposed operand tile: [rows][GB_K + 8]; transposed: [GB_K][rows + 8]
    // padding: +16 B per k-contiguous row (b128 fragment reads of 16 consecutive rows rotate through the banks); +64 B per
    // transposed row (the 64-byte pieces that 4 consecutive k-rows feed to one ds_read_b64_tr_b16 must not share banks)
    static constexpr int A_LD = AT ? (BM + 32) : (GB_K + 8);
    static constexpr int B_LD = BT ? (BN + 32) : (GB_K + 8);
    static constexpr int A_ELEMS = AT ? GB_K * A_LD : BM * A_LD;
    static constexpr int B_ELEMS = BT ? GB_K * B_LD : BN * B_LD;
    static constexpr int STAGE = A_ELEMS + B_ELEMS;
    static constexpr size_t PIPE_BYTES = (size_t)2 * STAGE * sizeof(bf16_t);
    static constexpr size_t EPI_BYTES = (size_t)BM * (BN + 4) * sizeof(float);   // fp32 epilogue tile (the bf16 one is smaller)
    static constexpr size_t BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
};

// XCD-aware workgroup -> tile mapping (guide T1, bijective form): hardware deals workgroups round-robin over the 8 XCDs, each
// with a private 4 MiB L2; remapping the linear id so that one XCD receives a CONTIGUOUS run of tiles (same row panel,
// consecutive column tiles, then the next row panel) keeps operand panels hot in that XCD's L2 instead of fetching every panel
// into all eight (rocprofv3 FETCH_SIZE showed 3.5x the algorithmic bytes on the weight-gradient GEMMs without it).
__device__ __forceinline__ void tile_coords(int nx, int ny, int nz, int &tx, int &ty, int &tz) {
    const int nwg = nx * ny * nz, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    tx = id % nx;
    ty = (id / nx) % ny;
    tz = id / (nx * ny);
}

// one 16-byte chunk (8 bf16) of an operand tile per (thread, iteration): global -> registers
template <int ROWS, bool TR>
struct TileLoader {
    // TR=false: tile is [ROWS][GB_K] of a [rows, K] matrix (ld = leading dim); TR=true: tile is [GB_K][ROWS] of a [K, rows] matrix
    static constexpr int CHUNKS = ROWS * GB_K / 8;
    static constexpr int ITERS = (CHUNKS + 255) / 256;
    uint4 v[ITERS];
    // every piece is ALWAYS requested (row / k clamped into the matrix) and masked afterwards: a guarded load sits in its own basic
    // block and is waited for on the spot, which turned the ITERS pieces of a k-tile into ITERS serialized memory round trips.
    // Rows beyond the matrix may hold anything (their results are never stored); k beyond K must read as zero.
    __device__ __forceinline__ void load(const bf16_t *__restrict__ src, long long ld, int row0, int nrows, int k0, int K) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = min(it * 256 + (int)threadIdx.x, CHUNKS - 1);
            if (!TR) {
                const int rr = c / (GB_K / 8), kk = (c % (GB_K / 8)) * 8;
                v[it] = *reinterpret_cast<const uint4 *>(src + (long long)min(row0 + rr, nrows - 1) * ld + min(k0 + kk, K - 8));
                if (k0 + kk >= K) v[it] = make_uint4(0, 0, 0, 0);
            } else {
                const int kk = c / (ROWS / 8), rr = (c % (ROWS / 8)) * 8;
                v[it] = *reinterpret_cast<const uint4 *>(src + (long long)min(k0 + kk, K - 1) * ld + min(row0 + rr, nrows - 8));
                if (k0 + kk >= K) v[it] = make_uint4(0, 0, 0, 0);
            }
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

// accumulators -> LDS tile (row-major, padded) -> 16-byte coalesced row stores (the register layout alone would give 2- or
// 4-byte stores strided by a row: store-issue bound at the model's short K). Caller guarantees all k-tile LDS reads are done.
template <int BM, int BN, int OUT_MODE>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[BM / 64][BN / 64], char *smem, void *__restrict__ Cv, int M, int N,
                                              long long ldc, long long slab_stride, int m0, int n0, int wm, int wn, int lane,
                                              const EpiArgs &ep, int tile_y, int tile_z) {
    constexpr int RB = BM / 64, CB = BN / 64;
    bf16_t *lds = reinterpret_cast<bf16_t *>(smem);
    if (OUT_MODE == 0 && ep.mode == 1) {
        // bias + LeakyReLU + dropout applied to the accumulators IN REGISTERS (accumulator layout: lane = one column, 16 rows per
        // 32x32 block), then the plain bf16 staging / 16-byte store path below: half the LDS traffic of the fp32-tile epilogue and
        // no second pass over the tile. An element's keep-bit is the half of its column pair's hash word selected by the column
        // parity (common.h): lanes r and r^1 evaluate the same hash - redundant but branch-free.
        const int r = lane & 31, hh = lane >> 5;
        unsigned long long seed = ep.seed;
        if (ep.seed_dev) seed += *ep.seed_dev;
        const unsigned thr = drop_thr16(ep.p);
        const float ks = drop_scale16(thr);
        const DropKey dk = drop_key(seed);
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int n = n0 + wn * (BN / 2) + 32 * j + r;
            const float bj = (ep.bias && n < N) ? ep.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int m = m0 + wm * (BM / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh;
                    float t = acc[i][j][g] + bj;
                    if (ep.slope >= 0.f) t = lrelu(t, ep.slope);
                    if (ep.p > 0.f) t = drop_keep1((unsigned long long)m * N + n, dk, thr) ? t * ks : 0.f;
                    acc[i][j][g] = t;
                }
        }
    } else if (OUT_MODE == 0 && ep.mode != 0) {   // fused elementwise epilogue: fp32 tile, one rounding at the end
        const int r = lane & 31, hh = lane >> 5;
        constexpr int LDT = BN + 4;
        constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
        constexpr int NIT = BM * CPR / 256;         // chunks per thread; 256 % CPR == 0, so a thread keeps ONE column group
        float *tile = reinterpret_cast<float *>(smem);
        bf16_t *Cb = reinterpret_cast<bf16_t *>(Cv);
        const int cc = (threadIdx.x % CPR) * 8, rr0 = threadIdx.x / CPR, n = n0 + cc;
        const bool n_ok = n + 8 <= N;
        // every global operand of the epilogue is requested BEFORE the accumulators go through LDS: a load issued inside the
        // store loop put one L2 round trip on each of its NIT iterations (+26 us on the 8000x2048x256 FFN GEMM)
        float bias8[8];
        uint4 yraw[NIT];
        if (ep.mode == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = (ep.bias && n_ok) ? ep.bias[n + e] : 0.f;
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = m0 + rr0 + it * (256 / CPR);
                // always issued (row / column clamped into the matrix): rows beyond M are never stored, so their value is irrelevant
                yraw[it] = *reinterpret_cast<const uint4 *>(ep.y + (long long)min(m, M - 1) * ep.ldy + min(n, N - 8));
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    tile[(wm * (BM / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + wn * (BN / 2) + 32 * j + r] = acc[i][j][g];
        __syncthreads();
        unsigned long long seed = ep.seed;
        if (ep.seed_dev) seed += *ep.seed_dev;
        const unsigned thr = drop_thr16(ep.p);
        const float ks = drop_scale16(thr);
        const DropKey dk = drop_key(seed);
        float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rr0 + it * (256 / CPR);
            const int m = m0 + rr;
            if (m < M && n_ok) {
                const float4 v_lo = *reinterpret_cast<const float4 *>(tile + rr * LDT + cc), v_hi = *reinterpret_cast<const float4 *>(tile + rr * LDT + cc + 4);
                float v[8] = {v_lo.x, v_lo.y, v_lo.z, v_lo.w, v_hi.x, v_hi.y, v_hi.z, v_hi.w};
                const unsigned long long idx = (unsigned long long)m * N + n;   // even: n % 8 == 0 and N % 8 == 0 on this path
                const unsigned km = ep.p > 0.f ? drop_keep_mask<8>(idx, dk, thr) : ~0u;
                if (ep.mode == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = v[e] + bias8[e];
                        if (ep.slope >= 0.f) t = lrelu(t, ep.slope);
                        if (ep.p > 0.f) t = ((km >> e) & 1u) ? t * ks : 0.f;
                        v[e] = t;
                    }
                } else {
                    const unsigned yw[4] = {yraw[it].x, yraw[it].y, yraw[it].z, yraw[it].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool y_neg = (yw[e >> 1] >> ((e & 1) ? 31 : 15)) & 1u;   // sign bit of the bf16 activation
                        const bool y_nz = ((yw[e >> 1] >> ((e & 1) ? 16 : 0)) & 0x7fffu) != 0;
                        float t = v[e];
                        if (ep.p > 0.f) t = ((km >> e) & 1u) ? t * ks : 0.f;
                        if (ep.slope >= 0.f && y_neg && y_nz) t *= ep.slope;
                        v[e] = t;
                        csum[e] += t;
                    }
                }
                st8(Cb + (long long)m * ldc + n, v);
            }
        }
        if (ep.mode == 2 && ep.colpart) {   // column sums of this tile (dbias partials): per-thread partials -> LDS -> one row of colpart
            __syncthreads();
            float *red = tile;               // [256 / CPR][BN + 4]
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rr0 * LDT + cc + e] = csum[e];
            __syncthreads();
            for (int t = threadIdx.x; t < BN; t += 256) {
                float sum = 0.f;
#pragma unroll 4
                for (int q = 0; q < 256 / CPR; ++q) sum += red[q * LDT + t];
                if (n0 + t < N) ep.colpart[(long long)tile_y * N + n0 + t] = sum;
            }
        }
        return;
    }
    const int r = lane & 31, hh = lane >> 5;
    if (OUT_MODE == 0) {
        constexpr int LDT = BN + 8;
        bf16_t *tile = lds;  // BM x LDT bf16 <= smem (all k-tile reads are behind the loop's last barrier)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    tile[(wm * (BM / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + wn * (BN / 2) + 32 * j + r] = (bf16_t)acc[i][j][g];
        __syncthreads();
        bf16_t *Cb = reinterpret_cast<bf16_t *>(Cv);
        const bool vec = (ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
        if (vec && m0 + BM <= M && n0 + BN <= N) {
            // a tile inside the matrix (all but the last row / column panel) stores without guards: behind the per-chunk guards below
            // every 16-byte store was waited for before the next one was issued (BM * BN / 2048 serialized round trips per thread)
#pragma unroll
            for (int it = 0; it < BM * (BN / 8) / 256; ++it) {
                const int c = threadIdx.x + it * 256, rr = c / (BN / 8), cc = (c % (BN / 8)) * 8;
                *reinterpret_cast<uint4 *>(Cb + (long long)(m0 + rr) * ldc + n0 + cc) = *reinterpret_cast<const uint4 *>(tile + rr * LDT + cc);
            }
            return;
        }
        for (int c = threadIdx.x; c < BM * (BN / 8); c += 256) {
            const int rr = c / (BN / 8), cc = (c % (BN / 8)) * 8;
            const int m = m0 + rr, n = n0 + cc;
            if (m >= M || n >= N) continue;
            if (vec && n + 8 <= N) *reinterpret_cast<uint4 *>(Cb + (long long)m * ldc + n) = *reinterpret_cast<const uint4 *>(tile + rr * LDT + cc);
            else
                for (int e = 0; e < 8 && n + e < N; ++e) Cb[(long long)m * ldc + n + e] = tile[rr * LDT + cc + e];
        }
    } else {
        constexpr int LDT = BN + 4;
        float *tile = reinterpret_cast<float *>(smem);  // BM x LDT fp32 <= smem for both tile sizes
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    tile[(wm * (BM / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + wn * (BN / 2) + 32 * j + r] = acc[i][j][g];
        __syncthreads();
        float *Cf = reinterpret_cast<float *>(Cv) + (OUT_MODE == 1 ? (long long)tile_z * slab_stride : 0);
        const bool vec = (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cf) & 15) == 0);
        if (vec && m0 + BM <= M && n0 + BN <= N) {      // unguarded form (see the bf16 path); OUT_MODE 2 requests all its reads first
            constexpr int NITF = BM * (BN / 4) / 256;
            float4 old[OUT_MODE == 2 ? NITF : 1];
            if (OUT_MODE == 2) {
#pragma unroll
                for (int it = 0; it < NITF; ++it) {
                    const int c = threadIdx.x + it * 256, rr = c / (BN / 4), cc = (c % (BN / 4)) * 4;
                    old[it] = *reinterpret_cast<const float4 *>(Cf + (long long)(m0 + rr) * ldc + n0 + cc);
                }
            }
#pragma unroll
            for (int it = 0; it < NITF; ++it) {
                const int c = threadIdx.x + it * 256, rr = c / (BN / 4), cc = (c % (BN / 4)) * 4;
                float4 v = *reinterpret_cast<const float4 *>(tile + rr * LDT + cc);
                if (OUT_MODE == 2) { v.x += old[it].x; v.y += old[it].y; v.z += old[it].z; v.w += old[it].w; }
                *reinterpret_cast<float4 *>(Cf + (long long)(m0 + rr) * ldc + n0 + cc) = v;
            }
            return;
        }
        for (int c = threadIdx.x; c < BM * (BN / 4); c += 256) {
            const int rr = c / (BN / 4), cc = (c % (BN / 4)) * 4;
            const int m = m0 + rr, n = n0 + cc;
            if (m >= M || n >= N) continue;
            float *dst = Cf + (long long)m * ldc + n;
            const float4 v = *reinterpret_cast<const float4 *>(tile + rr * LDT + cc);
            if (vec && n + 4 <= N) {
                if (OUT_MODE == 2) {
                    float4 o = *reinterpret_cast<float4 *>(dst);
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                    *reinterpret_cast<float4 *>(dst) = o;
                } else *reinterpret_cast<float4 *>(dst) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < 4 && n + e < N; ++e) {
                    if (OUT_MODE == 2) dst[e] += vv[e];
                    else dst[e] = vv[e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2 main loop: 3-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging), counted vmcnt + ONE raw
// s_barrier per k-tile, two tiles in flight behind the MFMAs. The DMA writes 1 KiB per wave-instruction linearly
// (base + lane*16), so rows are unpadded and bank conflicts are removed by a XOR swizzle applied to the per-lane SOURCE
// address and to the fragment reads alike (guide 5.4 rule 21). Requires K % 64 == 0 (no zero-fill path in a DMA).
// ---------------------------------------------------------------------------------------------------------------------
template <int ROWS, bool TR>
struct RingTile {
    static constexpr int ROW_BYTES = TR ? ROWS * 2 : GB_K * 2;          // bytes per LDS row
    static constexpr int NROWS = TR ? GB_K : ROWS;                       // LDS rows
    static constexpr int NCH = ROW_BYTES / 16;                           // 16-byte chunks per row
    static constexpr int BYTES = NROWS * ROW_BYTES;
    static constexpr int INSTR = BYTES / 4096;                           // wave-instructions per wave (4 waves x 1 KiB)
    // XOR swizzle of the 16-byte chunk index, applied to the DMA source address and to the fragment reads alike. Chosen per layout
    // so that every group of lanes the LDS serves together hits 64 distinct banks:
    //   k-contiguous tile (128-byte rows, ds_read_b128: 16 lanes = 16 consecutive rows, one chunk each): bank group =
    //   (row & 1) * 8 + chunk -> chunk ^ ((row >> 1) & 7) makes the 16 rows distinct (chunk ^ (row & 7) left rows r, r+8 colliding);
    //   transposed tile (rows of ROWS bf16, ds_read_b64_tr_b16: 32 lanes = 4 consecutive k-rows x 64 bytes): the 64-byte group
    //   (chunk >> 2) must differ between those rows -> 256-byte rows: chunk ^ ((row & 3) << 2); 128-byte rows (two groups per
    //   row, consecutive rows already alternate halves): chunk ^ (((row >> 1) & 1) << 2).
    __device__ static __forceinline__ int swz(int row, int ch) {
        if (!TR) return ch ^ ((row >> 1) & 7);
        return ROW_BYTES >= 256 ? ch ^ ((row & 3) << 2) : ch ^ (((row >> 1) & 1) << 2);
    }
    // issue this wave's share of one tile; rows/cols beyond the matrix are clamped (their results are never stored)
    __device__ static __forceinline__ void issue(const bf16_t *__restrict__ src, long long ld, int row0, int nrows, int k0, char *slot,
                                                 int wave, int lane) {
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            const int piece = i * 4 + wave;                              // 1 KiB piece index inside the tile
            const int byte = piece * 1024 + lane * 16;
            const int lrow = byte / ROW_BYTES, pos = (byte % ROW_BYTES) / 16;
            const int ch = swz(lrow, pos);                               // global chunk that lands in this LDS slot
            const bf16_t *g;
            if (!TR) g = src + (long long)min(row0 + lrow, nrows - 1) * ld + k0 + ch * 8;
            else g = src + (long long)(k0 + lrow) * ld + min(row0 + ch * 8, nrows - 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(slot + piece * 1024), 16, 0, 0);
        }
    }
    // ---- loop-invariant halves of issue() / frag(): everything that does not depend on the k-tile is computed ONCE per kernel.
    // (Left to the compiler, the transposed layouts re-derived swizzles, clamps and a 64-bit row*ld product for every DMA and every
    // fragment of every k-tile: ~70 VALU instructions per 4 MFMAs in the 64x64 weight-gradient kernel - VALU-bound.)
    static constexpr int NFO = TR ? 2 : 1;                               // LDS reads per fragment
    __device__ static __forceinline__ void src_ptrs(const bf16_t *__restrict__ src, long long ld, int row0, int nrows, int k0, int wave,
                                                    int lane, const bf16_t *(&g)[INSTR]) {
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            const int byte = (i * 4 + wave) * 1024 + lane * 16;
            const int lrow = byte / ROW_BYTES, pos = (byte % ROW_BYTES) / 16;
            const int ch = swz(lrow, pos);
            if (!TR) g[i] = src + (long long)min(row0 + lrow, nrows - 1) * ld + k0 + ch * 8;
            else g[i] = src + (long long)(k0 + lrow) * ld + min(row0 + ch * 8, nrows - 8);
        }
    }
    // elements to add to every source pointer per k-tile
    __device__ static __forceinline__ long long k_step(long long ld) { return TR ? (long long)GB_K * ld : (long long)GB_K; }
    __device__ static __forceinline__ void issue_at(const bf16_t *const (&g)[INSTR], long long off, char *slot, int wave) {
#pragma unroll
        for (int i = 0; i < INSTR; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g[i] + off),
                                             (__attribute__((address_space(3))) void *)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
    }
    __device__ static __forceinline__ void frag_offsets(int blk0, int s, int lane, int (&off)[NFO]) {
        const int r = lane & 31, hh = lane >> 5;
        if (!TR) {
            const int row = blk0 + r;
            off[0] = row * ROW_BYTES + swz(row, 2 * s + hh) * 16;
        } else {
            const int mhalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
            const int col = blk0 + 16 * mhalf + 4 * p4;
            const int row_lo = 16 * s + 8 * hh + q4, row_hi = row_lo + 4;
            off[0] = row_lo * ROW_BYTES + swz(row_lo, col >> 3) * 16 + (col & 7) * 2;
            off[NFO - 1] = row_hi * ROW_BYTES + swz(row_hi, col >> 3) * 16 + (col & 7) * 2;
        }
    }
    __device__ static __forceinline__ bf16x8 frag_at(const char *slot, const int (&off)[NFO]) {
        if (!TR) return *reinterpret_cast<const bf16x8 *>(slot + off[0]);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(slot + off[0]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(slot + off[NFO - 1]));
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    }
    __device__ static __forceinline__ bf16x8 frag(const char *slot, int blk0, int s, int lane) {
        const int r = lane & 31, hh = lane >> 5;
        if (!TR) {
            const int row = blk0 + r;
            return *reinterpret_cast<const bf16x8 *>(slot + row * ROW_BYTES + swz(row, 2 * s + hh) * 16);
        }
        const int mhalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int col = blk0 + 16 * mhalf + 4 * p4;                      // element column of this lane's 8-byte piece
        const int row_lo = 16 * s + 8 * hh + q4, row_hi = row_lo + 4;
        const char *a_lo = slot + row_lo * ROW_BYTES + swz(row_lo, col >> 3) * 16 + (col & 7) * 2;
        const char *a_hi = slot + row_hi * ROW_BYTES + swz(row_hi, col >> 3) * 16 + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a_lo));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(a_hi));
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    }
};

#define RING_STAGES 3
#define RING_THREADS 512
// WAVE SPECIALISATION (round 4): every ring kernel runs 8 waves - waves 0-3 compute (fragment reads + MFMA + the epilogue), waves 4-7 do
// nothing but issue the LDS-DMA ring. A wave that issues a DMA into a full vector-memory queue sits in the issue stage until older
// requests drain, and with one stream per wave the MFMAs behind that DMA in program order wait with it: lab floors of the 4-wave form at
// 8000 x 256 x 2048 (tools/gemm_bench.py --nn128, profiles/r04_notes.md): DMA ring alone 11.4 us, fragment reads + MFMA alone 8.8 -
// 10.9 us, together 16.1 - 17.4 us whatever the compute part costs; with loader waves 13.0 us (K = 768: 8.7 -> 6.9, K = 256: 5.5 -> 4.8).
// Protocol per k-tile: loaders wait (counted vmcnt) until their pieces of tile kt have landed, ALL waves meet at one raw s_barrier (tile kt
// visible to the compute waves, which are done with tile kt - 1), loaders issue tile kt + STAGES - 1 into the slot of tile kt - 1.
// After the loop the loader waves end; s_barrier counts only waves that have not terminated, so the epilogue's barriers are among waves 0-3.
__device__ __forceinline__ void ring_wait_landed(int younger_tiles, int lpt) {
    // at most `younger_tiles` k-tiles' DMA instructions of this wave (lpt each, lpt <= 8) may still be outstanding
    const int n = younger_tiles * lpt;
    if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the loader waves' whole life: fill the ring for k-tiles [0, nk)
template <typename TA, typename TB, int ST = RING_STAGES>
__device__ __forceinline__ void ring_loader(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, long long lda, long long ldb, int m0, int M,
                                            int n0, int N, int kbeg, int nk, char *smem, int cw, int lane) {
    constexpr int SLOT = TA::BYTES + TB::BYTES;
    constexpr int LPT = TA::INSTR + TB::INSTR;
    static_assert(LPT <= 8, "ring_wait_landed covers up to 8 DMA instructions per wave and k-tile");
    const bf16_t *ga[TA::INSTR], *gb[TB::INSTR];
    TA::src_ptrs(A, lda, m0, M, kbeg, cw, lane, ga);
    TB::src_ptrs(B, ldb, n0, N, kbeg, cw, lane, gb);
    const long long a_step = TA::k_step(lda), b_step = TB::k_step(ldb);
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
        if (t < nk) {
            TA::issue_at(ga, t * a_step, smem + t * SLOT, cw);
            TB::issue_at(gb, t * b_step, smem + t * SLOT + TA::BYTES, cw);
        }
    for (int kt = 0; kt < nk; ++kt) {
        ring_wait_landed(min(ST - 2, nk - 1 - kt), LPT);
        __builtin_amdgcn_s_barrier();
        if (kt + ST - 1 < nk) {
            char *slot = smem + ((kt + ST - 1) % ST) * SLOT;
            TA::issue_at(ga, (kt + ST - 1) * a_step, slot, cw);
            TB::issue_at(gb, (kt + ST - 1) * b_step, slot + TA::BYTES, cw);
        }
    }
}

template <int BM, int BN, bool AT, bool BT, int OUT_MODE, int ST = RING_STAGES>
__global__ __launch_bounds__(RING_THREADS, 1) void gemm_bf16_ring_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, void *__restrict__ Cv,
                                                                int M, int N, int K, long long lda, long long ldb, long long ldc, int kchunk,
                                                                long long slab_stride, int nsplit, EpiArgs ep) {
    using TA = RingTile<BM, AT>;
    using TB = RingTile<BN, BT>;
    constexpr int SLOT = TA::BYTES + TB::BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RB = BM / 64, CB = BN / 64;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3, wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords((N + BN - 1) / BN, (M + BM - 1) / BM, nsplit, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int kbeg = tz * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg) / GB_K;
    const bf16_t *Bm = B;
    if (ep.btab) {      // workgroup-uniform
        Bm = reinterpret_cast<const bf16_t *>(ep.btab[blockIdx.y]);
        Cv = reinterpret_cast<bf16_t *>(Cv) + (long long)blockIdx.y * ep.c_batch;
    }
    if (wave8 >= 4) {
        ring_loader<TA, TB, ST>(A, Bm, lda, ldb, m0, M, n0, N, kbeg, nk, smem, wave, lane);
        __builtin_amdgcn_s_barrier();     // pairs with the compute waves' barrier in front of the epilogue
        return;
    }
    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = (f32x16){0};
    int a_off[RB][GB_K / 16][TA::NFO], b_off[CB][GB_K / 16][TB::NFO];
#pragma unroll
    for (int s = 0; s < GB_K / 16; ++s) {
#pragma unroll
        for (int i = 0; i < RB; ++i) TA::frag_offsets(wm * (BM / 2) + 32 * i, s, lane, a_off[i][s]);
#pragma unroll
        for (int j = 0; j < CB; ++j) TB::frag_offsets(wn * (BN / 2) + 32 * j, s, lane, b_off[j][s]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                 // the loaders' pieces of tile kt have landed
        const char *as = smem + (kt % ST) * SLOT, *bs = as + TA::BYTES;
        // every fragment read of the k-tile is in flight before the first MFMA needs one (one wave per SIMD computes: nothing else hides
        // the LDS latency; read-per-k16-step cost 1.4 us of the 14.4 at K = 2048)
        bf16x8 af[RB][GB_K / 16], bfr[CB][GB_K / 16];
#pragma unroll
        for (int s = 0; s < GB_K / 16; ++s) {
#pragma unroll
            for (int i = 0; i < RB; ++i) af[i][s] = TA::frag_at(as, a_off[i][s]);
#pragma unroll
            for (int j = 0; j < CB; ++j) bfr[j][s] = TB::frag_at(bs, b_off[j][s]);
        }
#pragma unroll
        for (int s = 0; s < GB_K / 16; ++s)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bfr[j][s], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all fragment reads done before the epilogue tile overwrites the ring (the loader waves end behind it)
    gemm_epilogue<BM, BN, OUT_MODE>(acc, smem, Cv, M, N, ldc, slab_stride, m0, n0, wm, wn, lane, ep, ty, tz);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient tile (both operands k-major, 64x64 output, fp32 out): the four waves split the K dimension INSIDE the k-tile
// (wave w owns k16 step w) and each accumulates the whole 64x64 tile, instead of each owning one 32x32 block over all four steps.
// Same MFMA count per wave, but a wave now reads 2 + 2 fragments for 4 MFMAs instead of 4 x (1 + 1): half the
// ds_read_b64_tr_b16 traffic, which is what bounds these kernels. The four partial tiles are combined through LDS in a fixed
// order ((w0 + w2) + (w1 + w3)) before the slab / accumulate store.
// ---------------------------------------------------------------------------------------------------------------------
template <int OUT_MODE>
__global__ __launch_bounds__(RING_THREADS, 1) void gemm_tt64_wavek_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, float *__restrict__ C,
                                                                 int M, int N, int K, long long lda, long long ldb, long long ldc, int kchunk,
                                                                 long long slab_stride, int nsplit) {
    using TA = RingTile<64, true>;
    using TB = RingTile<64, true>;
    constexpr int SLOT = TA::BYTES + TB::BYTES;
    static_assert(GB_K / 16 == 4, "one k16 step per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3;
    int tx, ty, tz;
    tile_coords((N + 63) / 64, (M + 63) / 64, nsplit, tx, ty, tz);
    const int m0 = ty * 64, n0 = tx * 64;
    const int kbeg = tz * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg) / GB_K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    if (wave8 >= 4) {      // loader waves (see gemm_bf16_ring_kernel)
        ring_loader<TA, TB>(A, B, lda, ldb, m0, M, n0, N, kbeg, nk, smem, wave, lane);
        __builtin_amdgcn_s_barrier();
        return;
    }
    int a_off[2][TA::NFO], b_off[2][TB::NFO];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        TA::frag_offsets(32 * i, wave, lane, a_off[i]);
        TB::frag_offsets(32 * i, wave, lane, b_off[i]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                 // the loaders' pieces of tile kt have landed
        const char *as = smem + (kt % RING_STAGES) * SLOT, *bs = as + TA::BYTES;
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = TA::frag_at(as, a_off[i]);
            bfr[i] = TB::frag_at(bs, b_off[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all fragment reads done before the partial tiles overwrite the ring (the loader waves end behind it)
    constexpr int LDT = 64 + 4;
    static_assert(2 * 64 * LDT * 4 <= RING_STAGES * SLOT, "two fp32 tiles fit the ring");
    float *t0 = reinterpret_cast<float *>(smem), *t1 = t0 + 64 * LDT;
    const int r = lane & 31, hh = lane >> 5;
    float *mine = (wave & 1) ? t1 : t0;
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) mine[(32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + 32 * j + r] = acc[i][j][g];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float *q = mine + (32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + 32 * j + r;
                    *q = acc[i][j][g] + *q;
                }
    }
    __syncthreads();
    float *Cf = C + (OUT_MODE == 1 ? (long long)tz * slab_stride : 0);
    const bool vec = (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cf) & 15) == 0);
    for (int c = threadIdx.x; c < 64 * 16; c += 256) {
        const int rr = c >> 4, cc = (c & 15) * 4;
        const int m = m0 + rr, n = n0 + cc;
        if (m >= M || n >= N) continue;
        float *dst = Cf + (long long)m * ldc + n;
        const float4 va = *reinterpret_cast<const float4 *>(t0 + rr * LDT + cc), vb = *reinterpret_cast<const float4 *>(t1 + rr * LDT + cc);
        const float vv[4] = {va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w};
        if (vec && n + 4 <= N) {
            float4 o = make_float4(vv[0], vv[1], vv[2], vv[3]);
            if (OUT_MODE == 2) {
                const float4 old = *reinterpret_cast<float4 *>(dst);
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *reinterpret_cast<float4 *>(dst) = o;
        } else {
            for (int e = 0; e < 4 && n + e < N; ++e) {
                if (OUT_MODE == 2) dst[e] += vv[e];
                else dst[e] = vv[e];
            }
        }
    }
}

// The same wave-K split for the plain forward / data-gradient 64x64 tile (both operands k-contiguous, bf16 output): the main loops
// of these kernels are bound by LDS bandwidth, not by MFMA or by the DMA latency (bytes through LDS per 32x32x16 MFMA: 2 KB of
// fragment reads + 1 KB of DMA writes with one 32x32 block per wave; 1 + 1 KB with a 64x64 block per wave).
__global__ __launch_bounds__(RING_THREADS, 1) void gemm_nn64_wavek_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, bf16_t *__restrict__ C,
                                                                 int M, int N, int K, long long lda, long long ldb, long long ldc, int kchunk,
                                                                 long long slab_stride, int nsplit) {
    using TA = RingTile<64, false>;
    using TB = RingTile<64, false>;
    constexpr int SLOT = TA::BYTES + TB::BYTES;
    static_assert(GB_K / 16 == 4, "one k16 step per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3;
    int tx, ty, tz;
    tile_coords((N + 63) / 64, (M + 63) / 64, nsplit, tx, ty, tz);
    const int m0 = ty * 64, n0 = tx * 64;
    const int kbeg = tz * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg) / GB_K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    if (wave8 >= 4) {      // loader waves (see gemm_bf16_ring_kernel)
        ring_loader<TA, TB>(A, B, lda, ldb, m0, M, n0, N, kbeg, nk, smem, wave, lane);
        __builtin_amdgcn_s_barrier();
        return;
    }
    int a_off[2][TA::NFO], b_off[2][TB::NFO];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        TA::frag_offsets(32 * i, wave, lane, a_off[i]);
        TB::frag_offsets(32 * i, wave, lane, b_off[i]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                 // the loaders' pieces of tile kt have landed
        const char *as = smem + (kt % RING_STAGES) * SLOT, *bs = as + TA::BYTES;
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = TA::frag_at(as, a_off[i]);
            bfr[i] = TB::frag_at(bs, b_off[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all fragment reads done before the partial tiles overwrite the ring (the loader waves end behind it)
    constexpr int LDT = 64 + 4;
    static_assert(2 * 64 * LDT * 4 <= RING_STAGES * SLOT, "two fp32 tiles fit the ring");
    float *t0 = reinterpret_cast<float *>(smem), *t1 = t0 + 64 * LDT;
    const int r = lane & 31, hh = lane >> 5;
    float *mine = (wave & 1) ? t1 : t0;
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) mine[(32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + 32 * j + r] = acc[i][j][g];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float *q = mine + (32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + 32 * j + r;
                    *q = acc[i][j][g] + *q;
                }
    }
    __syncthreads();
    const bool vec = (ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    for (int c = threadIdx.x; c < 64 * 8; c += 256) {
        const int rr = c >> 3, cc = (c & 7) * 8;
        const int m = m0 + rr, n = n0 + cc;
        if (m >= M || n >= N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const float4 va = *reinterpret_cast<const float4 *>(t0 + rr * LDT + cc + e), vb = *reinterpret_cast<const float4 *>(t1 + rr * LDT + cc + e);
            v[e] = va.x + vb.x; v[e + 1] = va.y + vb.y; v[e + 2] = va.z + vb.z; v[e + 3] = va.w + vb.w;
        }
        bf16_t *dst = C + (long long)m * ldc + n;
        if (vec && n + 8 <= N) st8(dst, v);
        else
            for (int e = 0; e < 8 && n + e < N; ++e) dst[e] = (bf16_t)v[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LAB ONLY (tools/gemm_bench.py --floors; never launched by the product path): the 128x64 tile of the step's N = 256 projections in its
// ROUND-3 form - four waves that each issue their share of the LDS-DMA ring AND compute - with two floor modes. They are the evidence
// behind the loader waves of the ring kernels above (profiles/r04_notes.md section 1), at 8000 x 256 x 2048, operands hot in L2:
//   FLOOR 0: the GEMM, 16.1 - 17.4 us;   FLOOR 1: the DMA ring and its barriers alone (no fragment reads, no MFMA): 11.4 us = 69 GB/s
//   per CU = what the L2 -> LDS path delivers to one workgroup per CU;   FLOOR 2: fragment reads + MFMA alone (no DMA): 10.9 us.
//   Neither floor explains 16-17 us; with the DMA issue moved to waves of its own the same tile runs in 13.0 us.
// ---------------------------------------------------------------------------------------------------------------------
template <int FLOOR>
__global__ __launch_bounds__(256, 1) void gemm_nn128x64_lab_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, bf16_t *__restrict__ C,
                                                                   int M, int N, int K, long long lda, long long ldb, long long ldc) {
    using TA = RingTile<128, false>;
    using TB = RingTile<64, false>;
    constexpr int STAGES = 4, SLOT = TA::BYTES + TB::BYTES, LPT = TA::INSTR + TB::INSTR, NS = GB_K / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords((N + 63) / 64, (M + 127) / 128, 1, tx, ty, tz);
    const int m0 = ty * 128, n0 = tx * 64;
    const int nk = K / GB_K;
    f32x16 acc[2];
    acc[0] = (f32x16){0}; acc[1] = (f32x16){0};
    const bf16_t *ga[TA::INSTR], *gb[TB::INSTR];
    TA::src_ptrs(A, lda, m0, M, 0, wave, lane, ga);
    TB::src_ptrs(B, ldb, n0, N, 0, wave, lane, gb);
    const long long a_step = TA::k_step(lda), b_step = TB::k_step(ldb);
    int a_off[2][NS][TA::NFO], b_off[NS][TB::NFO];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) TA::frag_offsets(wm * 64 + 32 * i, s, lane, a_off[i][s]);
        TB::frag_offsets(wn * 32, s, lane, b_off[s]);
    }
    if (FLOOR != 2) {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) {
                TA::issue_at(ga, t * a_step, smem + t * SLOT, wave);
                TB::issue_at(gb, t * b_step, smem + t * SLOT + TA::BYTES, wave);
            }
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (FLOOR != 2) ring_wait_landed(min(STAGES - 2, nk - 1 - kt), LPT);   // tile kt has landed
        __builtin_amdgcn_s_barrier();                 // every wave's pieces landed; every wave is done reading slot (kt-1) % STAGES
        if (FLOOR != 2 && kt + STAGES - 1 < nk) {
            char *slot = smem + ((kt + STAGES - 1) % STAGES) * SLOT;
            TA::issue_at(ga, (kt + STAGES - 1) * a_step, slot, wave);
            TB::issue_at(gb, (kt + STAGES - 1) * b_step, slot + TA::BYTES, wave);
        }
        if (FLOOR == 1) continue;
        const char *as = smem + (kt % STAGES) * SLOT, *bs = as + TA::BYTES;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 bfr = TB::frag_at(bs, b_off[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TA::frag_at(as, a_off[i][s]), bfr, acc[i], 0, 0, 0);
        }
    }
    __syncthreads();
    const int r = lane & 31, hh = lane >> 5;
    constexpr int LDT = 64 + 4;
    float *t0 = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) t0[(wm * 64 + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh) * LDT + wn * 32 + r] = acc[i][g];
    __syncthreads();
    for (int c = threadIdx.x; c < 128 * 8; c += 256) {
        const int rr = c >> 3, cc = (c & 7) * 8;
        const int m = m0 + rr, n = n0 + cc;
        if (m >= M || n + 8 > N) continue;
        float v[8];
        ld8(t0 + rr * LDT + cc, v);
        st8(C + (long long)m * ldc + n, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Front-end block 2 (3x3 stride-2 Conv2d over channels-last [B,T,F,Ci] + the 1x1 stride-2 residual conv; SB/lobes/models/convolution.py:
// 178-266, SB/nnet/CNN.py:629-711) as IMPLICIT GEMMs: the [P, 9*Ci] patch matrix (369 MB at configs[1]) is never built. The loader waves of
// the ring kernels compute, per LDS-DMA piece, which input pixel a patch row comes from (the padding rule of csrc/frontend.hip's
// src_index: reflect / causal-zero) and point the DMA there; rows that fall into zero padding read a zero line. Compute waves are the
// ring kernels' unchanged. Forward: y1 = patches . Wm^T + b1 and y2 = centre tap . W2^T + b2 from the SAME staged A tiles (the centre
// segment's k-tiles feed a second accumulator against W2, resident in LDS). Filter gradient: dWm[Co, 9Ci] = dY^T . patches with the
// patch rows gathered as the k-major operand. (The data gradient stays dA = dY . Wm + col2im: a stride-2 tap hits an input pixel 1, 2 or
// 4 times, plus reflection fix-ups - no single GEMM.)
// ---------------------------------------------------------------------------------------------------------------------
struct ConvGeom {
    int B, T, F, To, Fo, Ci, tmode, fmode;   // input [B,T,F,Ci]; output positions P = B*To*Fo; padding modes as csrc/frontend.hip
};
__device__ __attribute__((aligned(256))) bf16_t g_conv_zero_line[128];   // zero-initialised: what a patch row in zero padding reads

__device__ __forceinline__ int conv_src_index(int o, int k, int n, int mode) {   // = frontend.hip src_index
    if (mode == 1) { const int i = 2 * o + k - 2; return i < 0 ? -1 : i; }
    int i = 2 * o + k - 1;
    if (mode == 0) { if (i < 0) i = -i; if (i >= n) i = 2 * (n - 1) - i; return i; }
    return (i < 0 || i >= n) ? -1 : i;
}
// output position (b, to, fo) of patch row p - 32-bit arithmetic (P < 2^31 is checked by the launchers): a 64-bit division per DMA piece
// and k-tile made the loader waves the bottleneck of the first version (conv_s2_wgrad 550 us against 290 us for the im2col + GEMM path)
struct ConvPos { int b, to, fo; };
__device__ __forceinline__ ConvPos conv_pos(const ConvGeom &g, unsigned p) {
    const unsigned q = p / (unsigned)g.Fo;
    return ConvPos{(int)(q / (unsigned)g.To), (int)(q % (unsigned)g.To), (int)(p - q * (unsigned)g.Fo)};
}
// ... advanced by `rows` patch rows (rows / Fo and rows % Fo precomputed: dq, dr) - no division on the per-k-tile path
__device__ __forceinline__ void conv_pos_advance(const ConvGeom &g, ConvPos &c, int dq, int dr) {
    c.fo += dr;
    c.to += dq;
    if (c.fo >= g.Fo) { c.fo -= g.Fo; c.to += 1; }
    while (c.to >= g.To) { c.to -= g.To; c.b += 1; }
}
// channel vector of the patch row at position c, tap (kt, kf): pointer to x[b, ti, fi, 0] or NULL for zero padding
__device__ __forceinline__ const bf16_t *conv_row_ptr(const bf16_t *__restrict__ x, const ConvGeom &g, const ConvPos &c, int kt, int kf) {
    const int ti = conv_src_index(c.to, kt, g.T, g.tmode), fi = conv_src_index(c.fo, kf, g.F, g.fmode);
    if (ti < 0 || fi < 0) return nullptr;
    return x + ((long long)(c.b * g.T + ti) * g.F + fi) * g.Ci;
}

// forward: grid = cdiv(P, 128) workgroups of 8 waves; Co == 128 (one column tile), Ci % 64 == 0
__global__ __launch_bounds__(RING_THREADS, 1) void conv_s2_fwd_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ Wm /*[128][9Ci]*/,
                                                                      const float *__restrict__ b1, const bf16_t *__restrict__ W2 /*[128][Ci]*/,
                                                                      const float *__restrict__ b2, bf16_t *__restrict__ y1, bf16_t *__restrict__ y2,
                                                                      ConvGeom g, long long P, int centre) {
    using TA = RingTile<128, false>;
    using TB = RingTile<128, false>;
    constexpr int SLOT = TA::BYTES + TB::BYTES, NS = GB_K / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *w2s = smem + RING_STAGES * SLOT;                 // W2 as Ci / 64 k-tiles of TB::BYTES, resident
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3, wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.x * 128;
    const int tps = g.Ci / GB_K;                           // k-tiles per tap segment
    const int nk = 9 * tps, K = 9 * g.Ci;
    if (wave8 >= 4) {
        // ---- loader waves: B (filters) by the regular strided pieces, A by the patch-row gather, W2 once
        constexpr int LPT = TA::INSTR + TB::INSTR;
        const bf16_t *gb[TB::INSTR], *g2[TB::INSTR];
        TB::src_ptrs(Wm, K, 0, 128, 0, wave, lane, gb);
        TB::src_ptrs(W2, g.Ci, 0, 128, 0, wave, lane, g2);
        for (int t2 = 0; t2 < tps; ++t2) TB::issue_at(g2, (long long)t2 * GB_K, w2s + t2 * TB::BYTES, wave);
        ConvPos prow[TA::INSTR];      // this lane's four patch rows: fixed for the whole kernel, only the tap changes with the k-tile
        int chunk[TA::INSTR];
#pragma unroll
        for (int i = 0; i < TA::INSTR; ++i) {
            const int byte = (i * 4 + wave) * 1024 + lane * 16;
            const int lrow = byte / TA::ROW_BYTES, pos = (byte % TA::ROW_BYTES) / 16;
            prow[i] = conv_pos(g, (unsigned)min(m0 + lrow, P - 1));
            chunk[i] = TA::swz(lrow, pos) * 8;
        }
        auto issue = [&](int kt) {
            char *slot = smem + (kt % RING_STAGES) * SLOT;
            const int seg = kt / tps, koff = (kt - seg * tps) * GB_K, tap_t = seg / 3, tap_f = seg - 3 * tap_t;
#pragma unroll
            for (int i = 0; i < TA::INSTR; ++i) {
                const bf16_t *rp = conv_row_ptr(x, g, prow[i], tap_t, tap_f);
                const bf16_t *src = rp ? rp + koff + chunk[i] : g_conv_zero_line + chunk[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
            }
            TB::issue_at(gb, (long long)kt * GB_K, slot + TA::BYTES, wave);
        };
        for (int t = 0; t < RING_STAGES - 1; ++t)
            if (t < nk) issue(t);
        for (int kt = 0; kt < nk; ++kt) {
            ring_wait_landed(min(RING_STAGES - 2, nk - 1 - kt), LPT);     // (the W2 pieces were issued first: landed with tile 0)
            __builtin_amdgcn_s_barrier();
            if (kt + RING_STAGES - 1 < nk) issue(kt + RING_STAGES - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] = (f32x16){0}; acc2[i][j] = (f32x16){0}; }
    int a_off[2][NS][TA::NFO], b_off[2][NS][TB::NFO];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            TA::frag_offsets(wm * 64 + 32 * i, s, lane, a_off[i][s]);
            TB::frag_offsets(wn * 64 + 32 * i, s, lane, b_off[i][s]);
        }
    const int c_lo = centre * tps, c_hi = c_lo + tps;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        const char *as = smem + (kt % RING_STAGES) * SLOT, *bs = as + TA::BYTES;
        bf16x8 af[2][NS], bfr[2][NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][s] = TA::frag_at(as, a_off[i][s]);
                bfr[i][s] = TB::frag_at(bs, b_off[i][s]);
            }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bfr[j][s], acc[i][j], 0, 0, 0);
        if (kt >= c_lo && kt < c_hi) {      // workgroup-uniform: the centre tap's tiles also feed the 1x1 stride-2 conv
            const char *b2s = w2s + (kt - c_lo) * TB::BYTES;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                bf16x8 b2f[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b2f[j] = TB::frag_at(b2s, b_off[j][s]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], b2f[j], acc2[i][j], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // the loader waves end behind it; the ring becomes the output staging tile
    constexpr int LDT = 128 + 8;
    bf16_t *tile = reinterpret_cast<bf16_t *>(smem);      // [128][LDT] bf16 = 34 KB
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float *bias = which ? b2 : b1;
        bf16_t *out = which ? y2 : y1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bj = bias[wn * 64 + 32 * j + r];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 16; ++gq) {
                    const float v = (which ? acc2[i][j][gq] : acc[i][j][gq]) + bj;
                    tile[(wm * 64 + 32 * i + (gq & 3) + 8 * (gq >> 2) + 4 * hh) * LDT + wn * 64 + 32 * j + r] = (bf16_t)v;
                }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 128 * 16; c += 256) {
            const int rr = c >> 4, cc = (c & 15) * 8;
            if (m0 + rr < P) *reinterpret_cast<uint4 *>(out + (m0 + rr) * 128 + cc) = *reinterpret_cast<const uint4 *>(tile + rr * LDT + cc);
        }
        __syncthreads();
    }
}

// filter gradient: slab[z][m][n] = sum over the z-th chunk of patch rows p of G[p][m] * patches[p][n_base + n], m < 128 (two 64-row tiles),
// n < Nout (64-column tiles). grid = (Nout / 64) * 2 * splits workgroups; the slabs are added in fixed order by the caller.
__global__ __launch_bounds__(RING_THREADS, 1) void conv_s2_wgrad_kernel(const bf16_t *__restrict__ G /*[P][128]*/, const bf16_t *__restrict__ x,
                                                                        float *__restrict__ slab, ConvGeom g, long long P, int n_base, int Nout,
                                                                        int kchunk, int nsplit) {
    using TA = RingTile<64, true>;
    using TB = RingTile<64, true>;
    constexpr int SLOT = TA::BYTES + TB::BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3;
    int tx, ty, tz;
    tile_coords(Nout / 64, 2, nsplit, tx, ty, tz);
    const int m0 = ty * 64, n0 = tx * 64;
    const long long kbeg = (long long)tz * kchunk, kend = min(P, kbeg + (long long)kchunk);
    const int nk = (int)((kend - kbeg + GB_K - 1) / GB_K);      // the last k-tile may run past P: its rows are clamped and ZERO-weighted below
    const int ncol = n_base + n0, seg = ncol / g.Ci, coff = ncol - seg * g.Ci;
    if (wave8 >= 4) {
        constexpr int LPT = TA::INSTR + TB::INSTR;
        int lrow[TB::INSTR], chunk[TB::INSTR];
        ConvPos pos[TB::INSTR];       // position of this lane's patch rows in the NEXT k-tile to issue (tiles are issued in order: + 64 rows each)
        const int tap_t = seg / 3, tap_f = seg - 3 * tap_t, dq = GB_K / g.Fo, dr = GB_K - dq * g.Fo;
#pragma unroll
        for (int i = 0; i < TB::INSTR; ++i) {
            const int byte = (i * 4 + wave) * 1024 + lane * 16;
            lrow[i] = byte / TB::ROW_BYTES;
            chunk[i] = TB::swz(lrow[i], (byte % TB::ROW_BYTES) / 16) * 8;
            pos[i] = conv_pos(g, (unsigned)min(kbeg + lrow[i], P - 1));
        }
        auto issue = [&](int kt) {
            char *slot = smem + (kt % RING_STAGES) * SLOT;
            const long long k0 = kbeg + (long long)kt * GB_K;
#pragma unroll
            for (int i = 0; i < TA::INSTR; ++i) {      // dY rows (k-major operand): rows beyond P read the zero line (they must not contribute)
                const int byte = (i * 4 + wave) * 1024 + lane * 16, lr = byte / TA::ROW_BYTES, ch = TA::swz(lr, (byte % TA::ROW_BYTES) / 16) * 8;
                const long long p = k0 + lr;
                const bf16_t *src = p < kend ? G + p * 128 + m0 + ch : g_conv_zero_line + ch;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TB::INSTR; ++i) {
                const long long p = k0 + lrow[i];
                const bf16_t *rp = p < kend ? conv_row_ptr(x, g, pos[i], tap_t, tap_f) : nullptr;
                conv_pos_advance(g, pos[i], dq, dr);
                const bf16_t *src = rp ? rp + coff + chunk[i] : g_conv_zero_line + chunk[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + TA::BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
            }
        };
        for (int t = 0; t < RING_STAGES - 1; ++t)
            if (t < nk) issue(t);
        for (int kt = 0; kt < nk; ++kt) {
            ring_wait_landed(min(RING_STAGES - 2, nk - 1 - kt), LPT);
            __builtin_amdgcn_s_barrier();
            if (kt + RING_STAGES - 1 < nk) issue(kt + RING_STAGES - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    int a_off[2][TA::NFO], b_off[2][TB::NFO];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        TA::frag_offsets(32 * i, wave, lane, a_off[i]);
        TB::frag_offsets(32 * i, wave, lane, b_off[i]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        const char *as = smem + (kt % RING_STAGES) * SLOT, *bs = as + TA::BYTES;
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = TA::frag_at(as, a_off[i]);
            bfr[i] = TB::frag_at(bs, b_off[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int LDT = 64 + 4;
    float *t0 = reinterpret_cast<float *>(smem), *t1 = t0 + 64 * LDT;
    const int r = lane & 31, hh = lane >> 5;
    float *mine = (wave & 1) ? t1 : t0;
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int gq = 0; gq < 16; ++gq) mine[(32 * i + (gq & 3) + 8 * (gq >> 2) + 4 * hh) * LDT + 32 * j + r] = acc[i][j][gq];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int gq = 0; gq < 16; ++gq) {
                    float *q = mine + (32 * i + (gq & 3) + 8 * (gq >> 2) + 4 * hh) * LDT + 32 * j + r;
                    *q = acc[i][j][gq] + *q;
                }
    }
    __syncthreads();
    float *Cf = slab + (long long)tz * 128 * Nout;
    for (int c = threadIdx.x; c < 64 * 16; c += 256) {
        const int rr = c >> 4, cc = (c & 15) * 4;
        const float4 va = *reinterpret_cast<const float4 *>(t0 + rr * LDT + cc), vb = *reinterpret_cast<const float4 *>(t1 + rr * LDT + cc);
        *reinterpret_cast<float4 *>(Cf + (long long)(m0 + rr) * Nout + n0 + cc) = make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient of front-end block 2 (round 5): dx[b, ti, fi, :] = sum over the taps (kt, kf) and output positions (to, fo) with
// src(to, kt) = ti, src(fo, kf) = fi of dy1[b, to, fo, :] . W1[:, (kt, kf), :]  (+ dy2[b, ti/2, fi/2, :] . W2 at even ti, fi) - a gathered GEMM
// per CLASS of input pixels instead of dA = dy1 . Wm (a [P, 9 Ci] matrix in HBM) + the inverse gather fe_col2im. Which (tap, output
// position) pairs reach an input index depends only on its parity, except next to a reflected border: along each axis the host plan
// (conv_dgrad_axis) sorts the indices into stride-2 runs whose contributions are (k, o = run index + c) for a fixed list of (k, c) - taps
// whose o falls outside [0, No) read the zero line - and SINGLE indices with their explicit list (the reflected rows: index 1, and n - 2
// when n is odd). A 2-D class = (time run) x (frequency run); its rows (b, it, if) are tiled by 128, each tile contracts over
// (slots of the class) x 128 output channels, the 1x1 branch rides as one more slot wherever the centre tap is in the list. The plan also
// fixes the launch order of the tiles: sorted by the time position of their first row and dealt to the XCDs in contiguous runs, so that the
// classes that gather the same dy rows run side by side on the same L2.
// ---------------------------------------------------------------------------------------------------------------------
struct ConvDgradClass {      // 32 ints, read through scalar loads (workgroup-uniform)
    int rows, nt, nf, t0, tstep, ost, f0, fstep, osf, nst, nsf, res_st, res_sf, pad0, pad1, pad2;
    int kt[4], ct[4], kf[4], cf[4];
};
#define CONV_DGRAD_HDR 8     // plan header ints: [0] classes, [1] tiles (launch slots, a multiple of 8), [2] class table offset, [3] tile table offset

__global__ __launch_bounds__(RING_THREADS, 1) void conv_s2_dgrad_kernel(const bf16_t *__restrict__ dy1, const bf16_t *__restrict__ dy2,
                                                                        const bf16_t *__restrict__ Wm /*[128][9Ci]*/, const bf16_t *__restrict__ W2 /*[128][Ci]*/,
                                                                        bf16_t *__restrict__ dx, ConvGeom g, const int *__restrict__ plan) {
    using TA = RingTile<128, false>;
    using TB = RingTile<64, true>;
    constexpr int SLOT = TA::BYTES + TB::BYTES, NS = GB_K / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3, wm = wave >> 1, wn = wave & 1;
    const int cls = plan[plan[3] + 2 * blockIdx.x], tile = plan[plan[3] + 2 * blockIdx.x + 1];
    if (cls < 0) return;      // a padding slot of the launch order (workgroup-uniform, before any barrier)
    const ConvDgradClass &c = *reinterpret_cast<const ConvDgradClass *>(plan + plan[2] + 32 * cls);
    const int n0 = blockIdx.y * 64, r0 = tile * 128;
    const int nreg = c.nst * c.nsf, nk = 2 * (nreg + (c.res_st >= 0 ? 1 : 0));
    if (wave8 >= 4) {
        constexpr int LPT = TA::INSTR + TB::INSTR;
        const bf16_t *gb[TB::INSTR], *g2[TB::INSTR];
        TB::src_ptrs(Wm, 9 * g.Ci, n0, g.Ci, 0, wave, lane, gb);
        TB::src_ptrs(W2, g.Ci, n0, g.Ci, 0, wave, lane, g2);
        int rb[TA::INSTR], rit[TA::INSTR], rif[TA::INSTR], chunk[TA::INSTR];      // this lane's four pixels: fixed, only the slot changes
#pragma unroll
        for (int i = 0; i < TA::INSTR; ++i) {
            const int byte = (i * 4 + wave) * 1024 + lane * 16;
            const int lrow = byte / TA::ROW_BYTES, pos = (byte % TA::ROW_BYTES) / 16;
            const unsigned row = (unsigned)min(r0 + lrow, c.rows - 1), q = row / (unsigned)c.nf;
            rif[i] = (int)(row - q * (unsigned)c.nf);
            rb[i] = (int)(q / (unsigned)c.nt);
            rit[i] = (int)(q - (unsigned)rb[i] * (unsigned)c.nt);
            chunk[i] = TA::swz(lrow, pos) * 8;
        }
        auto issue = [&](int kt) {
            char *slot = smem + (kt % RING_STAGES) * SLOT;
            const int sl = kt >> 1, half = kt & 1;
            const bool res = sl >= nreg;
            const int st = res ? c.res_st : sl / c.nsf, sf = res ? c.res_sf : sl - st * c.nsf;
            const int ct = c.ct[st], cf = c.cf[sf];
            const bf16_t *src = (res ? dy2 : dy1) + half * GB_K;
#pragma unroll
            for (int i = 0; i < TA::INSTR; ++i) {
                const int to = rit[i] * c.ost + ct, fo = rif[i] * c.osf + cf;
                const bool ok = (unsigned)to < (unsigned)g.To && (unsigned)fo < (unsigned)g.Fo;
                const bf16_t *p = ok ? src + ((long long)(rb[i] * g.To + to) * g.Fo + fo) * 128 + chunk[i] : g_conv_zero_line + chunk[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                 (__attribute__((address_space(3))) void *)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
            }
            if (res) TB::issue_at(g2, (long long)half * GB_K * g.Ci, slot + TA::BYTES, wave);
            else TB::issue_at(gb, (long long)half * GB_K * 9 * g.Ci + (c.kt[st] * 3 + c.kf[sf]) * g.Ci, slot + TA::BYTES, wave);
        };
        for (int t = 0; t < RING_STAGES - 1; ++t)
            if (t < nk) issue(t);
        for (int kt = 0; kt < nk; ++kt) {
            ring_wait_landed(min(RING_STAGES - 2, nk - 1 - kt), LPT);
            __builtin_amdgcn_s_barrier();
            if (kt + RING_STAGES - 1 < nk) issue(kt + RING_STAGES - 1);
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 acc[2];
    acc[0] = (f32x16){0};
    acc[1] = (f32x16){0};
    int a_off[2][NS][TA::NFO], b_off[NS][TB::NFO];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) TA::frag_offsets(wm * 64 + 32 * i, s, lane, a_off[i][s]);
        TB::frag_offsets(wn * 32, s, lane, b_off[s]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        const char *as = smem + (kt % RING_STAGES) * SLOT, *bs = as + TA::BYTES;
        bf16x8 af[2][NS], bfr[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][s] = TA::frag_at(as, a_off[i][s]);
            bfr[s] = TB::frag_at(bs, b_off[s]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bfr[s], acc[i], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // the loader waves end behind it; the ring becomes the output staging tile
    constexpr int LDT = 64 + 8;
    bf16_t *tl = reinterpret_cast<bf16_t *>(smem);      // [128][LDT] bf16
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) tl[(wm * 64 + 32 * i + (gq & 3) + 8 * (gq >> 2) + 4 * hh) * LDT + wn * 32 + r] = (bf16_t)acc[i][gq];
    __syncthreads();
    for (int e = threadIdx.x; e < 128 * 8; e += 256) {
        const int rr = e >> 3, cc = (e & 7) * 8;
        const int row = r0 + rr;
        if (row >= c.rows) continue;
        const unsigned q = (unsigned)row / (unsigned)c.nf, fi_i = (unsigned)row - q * (unsigned)c.nf, b = q / (unsigned)c.nt, ti_i = q - b * (unsigned)c.nt;
        const long long pix = ((long long)b * g.T + c.t0 + c.tstep * (int)ti_i) * g.F + c.f0 + c.fstep * (int)fi_i;
        *reinterpret_cast<uint4 *>(dx + pix * g.Ci + n0 + cc) = *reinterpret_cast<const uint4 *>(tl + rr * LDT + cc);
    }
}

// ---- host plan of conv_s2_dgrad_kernel ----
struct ConvAxisClass { int i0, step, n, ostep, ns, k[4], c[4]; };
static int conv_src_index_host(int o, int k, int n, int mode) {      // = conv_src_index
    if (mode == 1) { const int i = 2 * o + k - 2; return i < 0 ? -1 : i; }
    int i = 2 * o + k - 1;
    if (mode == 0) { if (i < 0) i = -i; if (i >= n) i = 2 * (n - 1) - i; return i; }
    return (i < 0 || i >= n) ? -1 : i;
}
// the classes of one axis (input length n, output length No, padding mode): false if an index collects more than four contributions
static bool conv_dgrad_axis(int n, int No, int mode, std::vector<ConvAxisClass> &out) {
    std::vector<std::vector<std::pair<int, int>>> L(n);
    for (int o = 0; o < No; ++o)
        for (int k = 0; k < 3; ++k) {
            const int i = conv_src_index_host(o, k, n, mode);
            if (i >= 0) L[i].push_back({k, o});
        }
    const int off = mode == 1 ? 2 : 1;
    std::vector<char> regular(n);
    for (int i = 0; i < n; ++i) {
        std::vector<std::pair<int, int>> e;
        for (int k = 0; k < 3; ++k)
            if (((i - k + off) & 1) == 0) {
                const int o = (i - k + off) / 2;
                if (i - k + off >= 0 && o < No) e.push_back({k, o});
            }
        std::sort(e.begin(), e.end());
        std::sort(L[i].begin(), L[i].end());
        regular[i] = e == L[i];
    }
    for (int i = 0; i < n; ++i) {
        if (regular[i]) {
            if (i >= 2 && regular[i - 2]) continue;      // inside a run that started earlier
            ConvAxisClass a{};
            a.i0 = i; a.step = 2; a.ostep = 1; a.n = 0;
            for (int j = i; j < n && regular[j]; j += 2) ++a.n;
            for (int k = 0; k < 3; ++k)
                if (((i - k + off) & 1) == 0) { a.k[a.ns] = k; a.c[a.ns] = (i - k + off) / 2; ++a.ns; }      // o of run index 0 (>= 0; the far end of the run may pass No - 1: zero line)
            out.push_back(a);
        } else {
            if (L[i].size() > 4) return false;
            ConvAxisClass a{};
            a.i0 = i; a.step = 0; a.ostep = 0; a.n = 1;
            for (auto &kc : L[i]) { a.k[a.ns] = kc.first; a.c[a.ns] = kc.second; ++a.ns; }
            if (a.ns == 0) { a.ns = 1; a.k[0] = 0; a.c[0] = -1; }      // an index nothing reads (cannot happen with stride 2, kernel 3): one slot of zeros
            out.push_back(a);
        }
    }
    return true;
}
struct ConvDgradPlan { std::vector<ConvDgradClass> cls; std::vector<int> tile_cls, tile_idx; int slots; };
static bool conv_dgrad_plan(int B, int T, int F, int causal, bool with_order, ConvDgradPlan &pl) {
    const int To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    std::vector<ConvAxisClass> ta, fa;
    if (!conv_dgrad_axis(T, To, causal ? 1 : 0, ta) || !conv_dgrad_axis(F, Fo, causal ? 2 : 0, fa)) return false;
    const int ckt = causal ? 2 : 1, ckf = 1;      // the tap that reads x[2 t', 2 f']
    long long tiles = 0;
    for (auto &a : ta)
        for (auto &f : fa) {
            ConvDgradClass c{};
            c.nt = a.n; c.nf = f.n; c.rows = B * a.n * f.n;
            c.t0 = a.i0; c.tstep = a.step; c.ost = a.ostep; c.f0 = f.i0; c.fstep = f.step; c.osf = f.ostep;
            c.nst = a.ns; c.nsf = f.ns; c.res_st = c.res_sf = -1;
            for (int s = 0; s < 4; ++s) { c.kt[s] = a.k[s]; c.ct[s] = a.c[s]; c.kf[s] = f.k[s]; c.cf[s] = f.c[s]; }
            for (int s = 0; s < a.ns; ++s)
                for (int q = 0; q < f.ns; ++q)
                    if (a.k[s] == ckt && f.k[q] == ckf) { c.res_st = s; c.res_sf = q; }
            pl.cls.push_back(c);
            tiles += (c.rows + 127) / 128;
        }
    if (tiles > (1 << 22)) return false;
    pl.slots = (int)((tiles + 7) / 8 * 8);
    if (!with_order) return true;
    // launch order: tiles sorted by (batch item, input time) of their first row, dealt to the 8 XCDs in contiguous runs (workgroup w runs on XCD w % 8)
    std::vector<std::pair<long long, std::pair<int, int>>> keyed;
    for (size_t ci = 0; ci < pl.cls.size(); ++ci) {
        const ConvDgradClass &c = pl.cls[ci];
        for (int t = 0; t < (c.rows + 127) / 128; ++t) {
            const long long r0 = (long long)t * 128, q = r0 / c.nf, b = q / c.nt, it = q % c.nt;
            keyed.push_back({b * T + c.t0 + c.tstep * it, {(int)ci, t}});
        }
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    pl.tile_cls.assign(pl.slots, -1);
    pl.tile_idx.assign(pl.slots, 0);
    const int per = pl.slots / 8;
    for (int w = 0; w < pl.slots; ++w) {
        const size_t s = (size_t)(w % 8) * per + w / 8;
        if (s < keyed.size()) { pl.tile_cls[w] = keyed[s].second.first; pl.tile_idx[w] = keyed[s].second.second; }
    }
    return true;
}

template <int BM, int BN, bool AT, bool BT>
struct RingSmem {
    static constexpr size_t PIPE = (size_t)RING_STAGES * (RingTile<BM, AT>::BYTES + RingTile<BN, BT>::BYTES);
    static constexpr size_t EPI = (size_t)BM * (BN + 4) * sizeof(float);
    static constexpr size_t BYTES = PIPE > EPI ? PIPE : EPI;
};

// OUT_MODE: 0 = bf16 store, 1 = fp32 store (slab or plain), 2 = fp32 accumulate (C += acc)
template <int BM, int BN, bool AT, bool BT, int OUT_MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, void *__restrict__ Cv,
                                                        int M, int N, int K, long long lda, long long ldb, long long ldc, int kchunk,
                                                        long long slab_stride, int nsplit, EpiArgs ep) {
    using S = GemmSmem<BM, BN, AT, BT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t *lds = reinterpret_cast<bf16_t *>(smem);
    constexpr int RB = BM / 64, CB = BN / 64;  // 32x32 blocks per wave in each direction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords((N + BN - 1) / BN, (M + BM - 1) / BM, nsplit, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int kbeg = tz * kchunk, kend = min(K, kbeg + kchunk);
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
    gemm_epilogue<BM, BN, OUT_MODE>(acc, smem, Cv, M, N, ldc, slab_stride, m0, n0, wm, wn, lane, ep, ty, tz);
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

// csrc/gemm_big.hip: 256-column output tiles, 8 waves (the feed-forward's wide GEMMs); returns 1 when the shape does not fit
int tsasr_gemm_big_bm(int M, int N, int K);
int tsasr_gemm_big_launch(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int mode,
                          const float *bias, const void *y, long long ldy, float slope, float p, unsigned long long seed,
                          const unsigned long long *seed_dev, float *colpart, void *mask, hipStream_t st);

static int g_use_ring = 1;   // 0: never, 1: long K or small tiles, 2: always
static int g_lab_floor = -1;   // tools/gemm_bench.py --floors: -1 = off; 0 / 1 / 2 = gemm_nn128x64_lab_kernel<FLOOR> for the 128x64 nn bf16 tile

template <int BM, int BN, bool AT, bool BT, int OUT_MODE>
static void launch(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int splits,
                   int kchunk, long long slab_stride, hipStream_t st, const EpiArgs &ep) {
    dim3 grid((unsigned)(cdiv(N, BN) * cdiv(M, BM) * splits));   // 1-D: tile coordinates come from the XCD-aware remap
    // measured (tools/gemm_bench.py, interleaved A/B): the DMA ring wins once a workgroup walks >= 16 k-tiles and
    // for the smaller macro-tiles at any K (3 workgroups per CU); the register-staged loop wins for 128x128 at short K (K = 256)
    const bool ring = g_use_ring && (K % GB_K == 0) && (kchunk % GB_K == 0) && (g_use_ring == 2 || min(K, kchunk) >= 1024 || BM * BN < 128 * 128) &&
                      (AT ? M >= 8 : true) && (BT ? N >= 8 : true);
    if constexpr (BM == 64 && BN == 64 && AT && BT && OUT_MODE != 0) {
        static const int wavek = 1;
        if (ring && wavek) {
            using R = RingSmem<64, 64, true, true>;
            gemm_tt64_wavek_kernel<OUT_MODE><<<grid, RING_THREADS, R::BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, (float *)C, M, N, K, lda, ldb, ldc, kchunk,
                                                                          slab_stride, splits);
            return;
        }
    }
    if constexpr (BM == 64 && BN == 64 && !AT && !BT && OUT_MODE == 0) {
        static const int wavek = 1;
        if (ring && wavek && ep.mode == 0 && min(K, kchunk) >= 16 * GB_K) {   // short K: the heavier epilogue costs more than the loop gains (5.4 -> 6.1 us at K = 256; 13.1 -> 11.2 us at M = 4000, K = 2048)
            using R = RingSmem<64, 64, false, false>;
            gemm_nn64_wavek_kernel<<<grid, RING_THREADS, R::BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, (bf16_t *)C, M, N, K, lda, ldb, ldc, kchunk,
                                                                slab_stride, splits);
            return;
        }
    }
    if constexpr (BM == 128 && BN == 64 && !AT && !BT && OUT_MODE == 0) {
        if (g_lab_floor >= 0 && ep.mode == 0 && !ep.btab && splits == 1 && K % GB_K == 0 && N % 8 == 0 && ldc % 8 == 0) {   // lab only
            constexpr int BYTES = 4 * (RingTile<128, false>::BYTES + RingTile<64, false>::BYTES);
            auto kern = g_lab_floor == 0 ? gemm_nn128x64_lab_kernel<0> : g_lab_floor == 1 ? gemm_nn128x64_lab_kernel<1> : gemm_nn128x64_lab_kernel<2>;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            kern<<<grid, 256, BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, (bf16_t *)C, M, N, K, lda, ldb, ldc);
            return;
        }
    }
    if (ring) {
        using R = RingSmem<BM, BN, AT, BT>;
        auto kern = gemm_bf16_ring_kernel<BM, BN, AT, BT, OUT_MODE>;
        if (R::BYTES > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::BYTES);
        kern<<<grid, RING_THREADS, R::BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, C, M, N, K, lda, ldb, ldc, kchunk, slab_stride, splits, ep);
        return;
    }
    using S = GemmSmem<BM, BN, AT, BT>;
    auto kern = gemm_bf16_kernel<BM, BN, AT, BT, OUT_MODE>;
    if (S::BYTES > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES);
    kern<<<grid, 256, S::BYTES, st>>>((const bf16_t *)A, (const bf16_t *)B, C, M, N, K, lda, ldb, ldc, kchunk, slab_stride, splits, ep);
}

template <int BM, int BN, int OUT_MODE>
static void launch_t(int tA, int tB, const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                     int splits, int kchunk, long long slab_stride, hipStream_t st, const EpiArgs &ep) {
    if (!tA && !tB) launch<BM, BN, false, false, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
    else if (!tA && tB) launch<BM, BN, false, true, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
    else if (tA && tB) launch<BM, BN, true, true, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
    else launch<BM, BN, true, false, OUT_MODE>(A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
}

struct GemmPlan { int tile; int splits; int kchunk; };   // tile: 0 = 128x128, 1 = 128x64, 2 = 64x64
static int g_force_tile = -1, g_force_splits = 0;             // tools/gemm_bench.py sweeps (tsasr_gemm_set_plan)
static GemmPlan plan(int M, int N, int K, int out_f32) {
    GemmPlan p;
    if (g_force_tile >= 0) {
        p.tile = g_force_tile;
        p.splits = (out_f32 && g_force_splits > 0) ? g_force_splits : 1;
        p.kchunk = cdiv(cdiv(K, p.splits), GB_K) * GB_K;
        p.splits = cdiv(K, p.kchunk);
        return p;
    }
    const long long t0 = (long long)cdiv(M, 128) * cdiv(N, 128), t1 = (long long)cdiv(M, 128) * cdiv(N, 64),
                    t2 = (long long)cdiv(M, 64) * cdiv(N, 64);
    // the largest macro-tile that still gives every CU work: 128x128 needs >= 2 tiles per CU (below that the 128x64 ring kernel
    // wins by 5-20% at N <= 768), 128x64 needs ~one per CU
    static const int thr0 = 512, thr1 = 192;
    p.tile = t0 >= thr0 ? 0 : (t1 >= thr1 ? 1 : 2);
    const long long tiles = p.tile == 0 ? t0 : (p.tile == 1 ? t1 : t2);
    p.splits = 1;
    static const int min_k = 384;   // (1024 -> 384: the K = 2T'-1 weight gradients of linear_pos ran on 16 workgroups; -0.12 ms per step)
    if (out_f32 && tiles < 256 && K >= min_k) {
        // weight gradients: few output tiles, long inner dimension -> split it (fp32 slabs, reduced in fixed order). Both operands
        // are k-major, i.e. every fragment comes through ds_read_b64_tr_b16, and that path - not the DMA, not VALU, not MFMA - is
        // what bounds these kernels (the same tiles read with ds_read_b128, wrong results, run in 21.5 us instead of 30.1 us).
        // Macro-tile / split sweeps stay within 10% of this choice in isolation and within noise in the step.
        // target: 3 workgroups per CU. Measured on the whole step (A/B in one process group, same box): 256 -> 19.0 ms, 384 -> 18.0,
        // 512 -> 17.65, 640 / 768 -> 17.4, 896 -> 17.85, 1024 -> 17.8 (more slabs = more bytes for the batched reduction).
        static const int target = 768;
        int s = (int)((target + tiles - 1) / tiles);
        const int max_s = std::max(1, K / (min_k >= 1024 ? 256 : 128));             // at least 4 (2 for short K) k-tiles per split
        if (s > max_s) s = max_s;
        if (s > 32) s = 32;
        if (s < 1) s = 1;
        p.splits = s;
    }
    p.kchunk = cdiv(cdiv(K, p.splits), GB_K) * GB_K;
    p.splits = cdiv(K, p.kchunk);
    return p;
}

template <int OUT_MODE>
static void launch_tile(int tile, int tA, int tB, const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb,
                        long long ldc, int splits, int kchunk, long long slab_stride, hipStream_t st, const EpiArgs &ep = EpiArgs{}) {
    if (tile == 0) launch_t<128, 128, OUT_MODE>(tA, tB, A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
    else if (tile == 1) launch_t<128, 64, OUT_MODE>(tA, tB, A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
    else launch_t<64, 64, OUT_MODE>(tA, tB, A, B, C, M, N, K, lda, ldb, ldc, splits, kchunk, slab_stride, st, ep);
}

static int tile_bm(int tile) { return tile == 2 ? 64 : 128; }

extern "C" {

/* 1 (default): LDS-DMA ring main loop for long inner dimensions; 2: ring whenever K % 64 == 0; 0: register-staged loop only (A/B tests). */
void tsasr_gemm_set_ring(int on) { g_use_ring = on; }
/* A/B tests only: force the macro-tile (0 = 128x128, 1 = 128x64, 2 = 64x64; -1 = automatic) and the split-K factor of fp32-output GEMMs. */
void tsasr_gemm_set_plan(int tile, int splits) { g_force_tile = tile; g_force_splits = splits; }
/* LAB (tools/gemm_bench.py --floors): -1 = off; 0 / 1 / 2: the 128x64 nn bf16 tile runs gemm_nn128x64_lab_kernel<floor> - the round-3 four-wave
 * loop / its LDS-DMA ring alone / its fragment reads + MFMA alone (modes 1 and 2 do NOT compute the product). */
void tsasr_gemm_set_lab_floor(int floor_mode) { g_lab_floor = floor_mode; }

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
    static const bool log_small = false;   // debugging aid: long-K problems that fill few CUs
    if (log_small && (long long)cdiv(M, tile_bm(p.tile)) * cdiv(N, p.tile == 0 ? 128 : 64) * p.splits <= 16)
        fprintf(stderr, "[tsasr_gemm_bf16] few tiles: M=%d N=%d K=%d tA=%d tB=%d out=%d acc=%d tile=%d splits=%d\n", M, N, K, transA, transB, out_dtype, accumulate, p.tile, p.splits);
    if (p.splits > 1) {
        TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_gemm_bf16_workspace_bytes(M, N, K, out_dtype), "tsasr_gemm_bf16: workspace too small");
        TSASR_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "tsasr_gemm_bf16: split-K output needs N, ldc multiples of 4");
        const long long ss = (long long)M * N;
        launch_tile<1>(p.tile, transA, transB, A, B, workspace, M, N, K, lda, ldb, N, p.splits, p.kchunk, ss, st);
        if (accumulate == 2 && ldc == N && ss < (1ll << 31) && tsasr_reduce_deferring())   // slabs stay in the workspace until tsasr_reduce_flush
            tsasr_reduce_submit((const float *)workspace, (float *)C, ss, p.splits, (int)ss, 1, st);
        else
            gemm_slab_reduce_kernel<<<(unsigned)cdiv((int)((ss + 3) / 4), 256), 256, 0, st>>>((const float *)workspace, (float *)C, M, N, ldc, p.splits, ss, accumulate != 0);
    } else if (out_dtype == TSASR_BF16) {
        if (!transA && !transB && g_force_tile < 0 && ldc % 8 == 0 &&
            tsasr_gemm_big_launch(A, B, C, M, N, K, lda, ldb, ldc, 0, nullptr, nullptr, 0, -1.f, 0.f, 0, nullptr, nullptr, nullptr, st) == 0) {
            TSASR_CHECK_LAUNCH("tsasr_gemm_bf16");
            return 0;
        }
        launch_tile<0>(p.tile, transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    } else if (accumulate) {
        launch_tile<2>(p.tile, transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    } else {
        launch_tile<1>(p.tile, transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, p.kchunk, 0, st);
    }
    TSASR_CHECK_LAUNCH("tsasr_gemm_bf16");
    return 0;
}

/* C_i [M,N] = A [M,K] . B_i^T for i < nbatch in ONE launch: A shared, B_i = btab[i] (DEVICE array of nbatch device pointers to bf16
 * [N,K] matrices, row stride ldb), C_i = C + i * c_batch elements (row stride ldc), all bf16. K % 64 == 0, N % 8 == 0. */
int tsasr_gemm_bf16_nt_batched(const void *A, const void *const *btab, void *C, int M, int N, int K, long long lda, long long ldb,
                               long long ldc, long long c_batch, int nbatch, void *stream) {
    TSASR_CHECK_ARG(A && btab && C && M > 0 && N > 0 && K > 0 && nbatch > 0 && nbatch < 65536, "tsasr_gemm_bf16_nt_batched: bad arguments");
    TSASR_CHECK_ARG(K % GB_K == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && c_batch % 8 == 0,
                    "tsasr_gemm_bf16_nt_batched: K must be a multiple of %d, N and the strides multiples of 8 bf16", GB_K);
    EpiArgs ep{};
    ep.btab = btab; ep.c_batch = c_batch;
    using R = RingSmem<64, 64, false, false>;
    dim3 grid((unsigned)(cdiv(N, 64) * cdiv(M, 64)), (unsigned)nbatch);
    gemm_bf16_ring_kernel<64, 64, false, false, 0><<<grid, RING_THREADS, R::BYTES, (hipStream_t)stream>>>((const bf16_t *)A, nullptr, C, M, N, K, lda, ldb, ldc, K, 0, 1, ep);
    TSASR_CHECK_LAUNCH("tsasr_gemm_bf16_nt_batched");
    return 0;
}

/* Front-end block 2 as implicit GEMMs (see conv_s2_fwd_kernel): x [B,T,F,Ci] bf16 channels-last; Wm [128, 9*Ci] = the 3x3 filter as
 * [Co, (kt, kf, ci)], W2 [128, Ci] the 1x1 filter (both bf16); b1 / b2 fp32 [128]; y1 = conv3x3_s2(x) + b1, y2 = conv1x1_s2(x) + b2, both
 * [B,T',F',128] bf16. causal: padding rule of the causal front-end (csrc/frontend.hip). Co == 128, Ci % 64 == 0, Ci <= 128. */
int tsasr_conv3x3s2_fwd(const void *x, const void *Wm, const float *b1, const void *W2, const float *b2, void *y1, void *y2, int B, int T, int F,
                        int Ci, int Co, int causal, void *stream) {
    TSASR_CHECK_ARG(x && Wm && b1 && W2 && b2 && y1 && y2, "tsasr_conv3x3s2_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && Co == 128 && Ci % 64 == 0 && Ci >= 64 && Ci <= 128, "tsasr_conv3x3s2_fwd: unsupported shape (Ci=%d Co=%d)", Ci, Co);
    ConvGeom g{B, T, F, (T - 1) / 2 + 1, (F - 1) / 2 + 1, Ci, causal ? 1 : 0, causal ? 2 : 0};
    const long long P = (long long)B * g.To * g.Fo;
    TSASR_CHECK_ARG(P < (1ll << 31) && (long long)B * T * F < (1ll << 31), "tsasr_conv3x3s2_fwd: more than 2^31 positions");
    const int lds = RING_STAGES * (RingTile<128, false>::BYTES * 2) + (Ci / GB_K) * RingTile<128, false>::BYTES;
    (void)hipFuncSetAttribute((const void *)conv_s2_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    conv_s2_fwd_kernel<<<(unsigned)((P + 127) / 128), RING_THREADS, lds, (hipStream_t)stream>>>((const bf16_t *)x, (const bf16_t *)Wm, b1, (const bf16_t *)W2, b2,
                                                                                               (bf16_t *)y1, (bf16_t *)y2, g, P, causal ? 7 : 4);
    TSASR_CHECK_LAUNCH("tsasr_conv3x3s2_fwd");
    return 0;
}

static int conv_wgrad_splits(long long P, int Nout) {
    const int tiles = (Nout / 64) * 2;
    int s = std::max(1, 768 / tiles);
    const long long max_s = std::max<long long>(1, P / (4 * GB_K));
    return (int)std::min<long long>(std::min<long long>(s, max_s), 64);
}
size_t tsasr_conv3x3s2_wgrad_workspace_bytes(int B, int T, int F, int Ci) {
    const long long P = (long long)B * ((T - 1) / 2 + 1) * ((F - 1) / 2 + 1);
    return align_up((size_t)conv_wgrad_splits(P, 9 * Ci) * 128 * 9 * Ci * sizeof(float), 256) +
           align_up((size_t)conv_wgrad_splits(P, Ci) * 128 * Ci * sizeof(float), 256);
}
/* dWm [128, 9*Ci] = dy1^T . patches(x), dW2 [128, Ci] = dy2^T . centre tap (fp32, OVERWRITTEN; the workspace holds the split-K slabs, added
 * in fixed order by two small launches). dy1, dy2 [P, 128] bf16. */
// dW1 [co][ci][kf][kt] = sum_z slab[z][co][(kt * 3 + kf) * Ci + ci]  (fixed order): the 3x3 filter gradient in the reference's parameter layout
__global__ __launch_bounds__(256) void conv_filter_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dW, int Ci, int nslab, long long slab_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // index in the slabs' order [co][(kt, kf)][ci]: coalesced reads of every slab, one scattered store
    if (i >= 128 * 9 * Ci) return;
    const int ci = i % Ci, tap = (i / Ci) % 9, co = i / (9 * Ci), kt = tap / 3, kf = tap % 3;
    float a = 0.f;
    for (int z = 0; z < nslab; ++z) a += slab[(long long)z * slab_stride + i];
    dW[((long long)co * Ci + ci) * 9 + kf * 3 + kt] = a;
}

static int conv3x3s2_wgrad_impl(const void *dy1, const void *dy2, const void *x, float *dWm, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                                void *workspace, size_t workspace_bytes, void *stream, bool filter_layout);

int tsasr_conv3x3s2_wgrad(const void *dy1, const void *dy2, const void *x, float *dWm, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                          void *workspace, size_t workspace_bytes, void *stream) {
    return conv3x3s2_wgrad_impl(dy1, dy2, x, dWm, dW2, B, T, F, Ci, Co, causal, workspace, workspace_bytes, stream, false);
}
/* The same, the 3x3 gradient written as the reference keeps the parameter: dW1 [128][Ci][kF = 3][kT = 3] (SB/nnet/CNN.py Conv2d weight) - the caller
 * adds it to the parameter's gradient as it is (no permuting copy). */
int tsasr_conv3x3s2_wgrad_filters(const void *dy1, const void *dy2, const void *x, float *dW1, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    return conv3x3s2_wgrad_impl(dy1, dy2, x, dW1, dW2, B, T, F, Ci, Co, causal, workspace, workspace_bytes, stream, true);
}

static int conv3x3s2_wgrad_impl(const void *dy1, const void *dy2, const void *x, float *dWm, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                                void *workspace, size_t workspace_bytes, void *stream, bool filter_layout) {
    TSASR_CHECK_ARG(dy1 && dy2 && x && dWm && dW2 && workspace, "tsasr_conv3x3s2_wgrad: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && Co == 128 && Ci % 64 == 0 && Ci >= 64 && Ci <= 128, "tsasr_conv3x3s2_wgrad: unsupported shape (Ci=%d Co=%d)", Ci, Co);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_conv3x3s2_wgrad_workspace_bytes(B, T, F, Ci), "tsasr_conv3x3s2_wgrad: workspace too small");
    ConvGeom g{B, T, F, (T - 1) / 2 + 1, (F - 1) / 2 + 1, Ci, causal ? 1 : 0, causal ? 2 : 0};
    const long long P = (long long)B * g.To * g.Fo;
    TSASR_CHECK_ARG(P < (1ll << 31) && (long long)B * T * F < (1ll << 31), "tsasr_conv3x3s2_wgrad: more than 2^31 positions");
    hipStream_t st = (hipStream_t)stream;
    constexpr int lds = RING_STAGES * 2 * RingTile<64, true>::BYTES;
    float *slab1 = (float *)workspace;
    const int s1 = conv_wgrad_splits(P, 9 * Ci), s2 = conv_wgrad_splits(P, Ci);
    float *slab2 = (float *)((char *)workspace + align_up((size_t)s1 * 128 * 9 * Ci * sizeof(float), 256));
    const int kc1 = (int)(((P + s1 - 1) / s1 + GB_K - 1) / GB_K * GB_K), kc2 = (int)(((P + s2 - 1) / s2 + GB_K - 1) / GB_K * GB_K);
    const int z1 = (int)((P + kc1 - 1) / kc1), z2 = (int)((P + kc2 - 1) / kc2);
    conv_s2_wgrad_kernel<<<(unsigned)((9 * Ci / 64) * 2 * z1), RING_THREADS, lds, st>>>((const bf16_t *)dy1, (const bf16_t *)x, slab1, g, P, 0, 9 * Ci, kc1, z1);
    conv_s2_wgrad_kernel<<<(unsigned)((Ci / 64) * 2 * z2), RING_THREADS, lds, st>>>((const bf16_t *)dy2, (const bf16_t *)x, slab2, g, P, (causal ? 7 : 4) * Ci, Ci, kc2, z2);
    // the split-K slabs are added right away, in fixed order (the caller permutes dWm into the reference's filter layout next)
    if (filter_layout) conv_filter_reduce_kernel<<<(unsigned)cdiv(128 * 9 * Ci, 256), 256, 0, st>>>(slab1, dWm, Ci, z1, (long long)128 * 9 * Ci);
    else gemm_slab_reduce_kernel<<<(unsigned)cdiv(128 * 9 * Ci / 4, 256), 256, 0, st>>>(slab1, dWm, 128, 9 * Ci, 9 * Ci, z1, (long long)128 * 9 * Ci, 0);
    gemm_slab_reduce_kernel<<<(unsigned)cdiv(128 * Ci / 4, 256), 256, 0, st>>>(slab2, dW2, 128, Ci, Ci, z2, (long long)128 * Ci, 0);
    TSASR_CHECK_LAUNCH("tsasr_conv3x3s2_wgrad");
    return 0;
}

/* Data gradient of the same block as ONE gathered GEMM launch (conv_s2_dgrad_kernel): dx [B,T,F,Ci] bf16 (every element written) from dy1,
 * dy2 [P,128] bf16 and the filters as the forward takes them. The plan (classes of input pixels + launch order of their tiles) depends on
 * the shape only: tsasr_conv3x3s2_dgrad_plan fills a HOST buffer of tsasr_conv3x3s2_dgrad_plan_bytes, the caller keeps a device copy. */
size_t tsasr_conv3x3s2_dgrad_plan_bytes(int B, int T, int F, int causal) {
    ConvDgradPlan pl;
    if (B <= 0 || T < 2 || F < 2 || !conv_dgrad_plan(B, T, F, causal, false, pl)) return 0;
    return (size_t)(CONV_DGRAD_HDR + 32 * pl.cls.size() + 2 * (size_t)pl.slots) * sizeof(int);
}
int tsasr_conv3x3s2_dgrad_plan(int B, int T, int F, int causal, void *plan_host, size_t plan_bytes) {
    TSASR_CHECK_ARG(plan_host && B > 0 && T >= 2 && F >= 2, "tsasr_conv3x3s2_dgrad_plan: bad argument");
    ConvDgradPlan pl;
    TSASR_CHECK_ARG(conv_dgrad_plan(B, T, F, causal, true, pl), "tsasr_conv3x3s2_dgrad_plan: unsupported shape (T=%d F=%d)", T, F);
    const size_t need = (size_t)(CONV_DGRAD_HDR + 32 * pl.cls.size() + 2 * (size_t)pl.slots) * sizeof(int);
    TSASR_CHECK_ARG(plan_bytes >= need, "tsasr_conv3x3s2_dgrad_plan: buffer too small");
    int *p = (int *)plan_host;
    memset(p, 0, need);
    p[0] = (int)pl.cls.size(); p[1] = pl.slots; p[2] = CONV_DGRAD_HDR; p[3] = CONV_DGRAD_HDR + 32 * (int)pl.cls.size();
    p[4] = B; p[5] = T; p[6] = F; p[7] = causal ? 1 : 0;
    static_assert(sizeof(ConvDgradClass) == 32 * sizeof(int), "class record = 32 ints");
    memcpy(p + p[2], pl.cls.data(), pl.cls.size() * sizeof(ConvDgradClass));
    for (int w = 0; w < pl.slots; ++w) { p[p[3] + 2 * w] = pl.tile_cls[w]; p[p[3] + 2 * w + 1] = pl.tile_idx[w]; }
    return 0;
}
int tsasr_conv3x3s2_dgrad(const void *dy1, const void *dy2, const void *Wm, const void *W2, void *dx, int B, int T, int F, int Ci, int Co, int causal,
                          const void *plan_dev, size_t plan_bytes, void *stream) {
    TSASR_CHECK_ARG(dy1 && dy2 && Wm && W2 && dx && plan_dev, "tsasr_conv3x3s2_dgrad: null pointer");
    TSASR_CHECK_ARG(B > 0 && T >= 2 && F >= 2 && Co == 128 && Ci % 64 == 0 && Ci >= 64 && Ci <= 128, "tsasr_conv3x3s2_dgrad: unsupported shape (Ci=%d Co=%d)", Ci, Co);
    ConvGeom g{B, T, F, (T - 1) / 2 + 1, (F - 1) / 2 + 1, Ci, causal ? 1 : 0, causal ? 2 : 0};
    TSASR_CHECK_ARG((long long)B * T * F < (1ll << 31), "tsasr_conv3x3s2_dgrad: more than 2^31 positions");
    ConvDgradPlan pl;
    TSASR_CHECK_ARG(conv_dgrad_plan(B, T, F, causal, false, pl), "tsasr_conv3x3s2_dgrad: unsupported shape (T=%d F=%d)", T, F);
    TSASR_CHECK_ARG(plan_bytes >= (size_t)(CONV_DGRAD_HDR + 32 * pl.cls.size() + 2 * (size_t)pl.slots) * sizeof(int), "tsasr_conv3x3s2_dgrad: plan of another shape");
    constexpr int lds = RING_STAGES * (RingTile<128, false>::BYTES + RingTile<64, true>::BYTES);
    (void)hipFuncSetAttribute((const void *)conv_s2_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    conv_s2_dgrad_kernel<<<dim3((unsigned)pl.slots, (unsigned)(Ci / 64)), RING_THREADS, lds, (hipStream_t)stream>>>(
        (const bf16_t *)dy1, (const bf16_t *)dy2, (const bf16_t *)Wm, (const bf16_t *)W2, (bf16_t *)dx, g, (const int *)plan_dev);
    TSASR_CHECK_LAUNCH("tsasr_conv3x3s2_dgrad");
    return 0;
}

size_t tsasr_gemm_bf16_fused_workspace_bytes(int M, int N) { return align_up((size_t)cdiv(M, 64) * N * sizeof(float), 256); }

/* bf16-output GEMM with a fused elementwise epilogue (no split-K):
 *   epi_mode 1: C = dropout_p(LeakyReLU_slope(op(A).op(B) + bias[n]))           (slope < 0: no activation; bias may be NULL)
 *   epi_mode 2: C = (op(A).op(B)) * keep(m,n)/(1-p) * LeakyReLU'(y[m,n]); dbias[n] = column sums of C (dbias may be NULL)
 * y [M,N] bf16 (row stride ldy) is the saved output of the mode-1 call with the same (p, seed): the mask is regenerated. */
/* 1 when the mode-1 AND mode-2 calls of this shape (transA = transB = 0) keep the epilogue mask words (see `mask` below) */
int tsasr_gemm_bf16_fused_mask_ok(int M, int N, int K) { return (g_force_tile < 0 && tsasr_gemm_big_bm(M, N, K)) ? 1 : 0; }

int tsasr_gemm_bf16_fused(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                          int transA, int transB, int epi_mode, const float *bias, const void *y, long long ldy, float slope, float p,
                          unsigned long long seed, const unsigned long long *seed_dev, float *dbias, void *mask, void *workspace,
                          size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(A && B && C, "tsasr_gemm_bf16_fused: null pointer");
    TSASR_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && ldc % 8 == 0, "tsasr_gemm_bf16_fused: N and ldc must be multiples of 8");
    TSASR_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && (transA ? M : K) % 8 == 0 && (transB ? N : K) % 8 == 0, "tsasr_gemm_bf16_fused: rows must be multiples of 8 bf16");
    TSASR_CHECK_ARG(epi_mode == 1 || (epi_mode == 2 && y && ldy % 8 == 0), "tsasr_gemm_bf16_fused: bad epilogue mode %d", epi_mode);
    TSASR_CHECK_ARG(!mask || (!transA && !transB && tsasr_gemm_bf16_fused_mask_ok(M, N, K)), "tsasr_gemm_bf16_fused: no mask words for this shape (ask tsasr_gemm_bf16_fused_mask_ok)");
    TSASR_CHECK_ARG(p >= 0.f && p < 1.f, "tsasr_gemm_bf16_fused: bad dropout %f", p);
    TSASR_CHECK_ARG(!(mask && slope > 1.f), "tsasr_gemm_bf16_fused: mask words need a LeakyReLU slope in [0, 1] (or none), got %f", slope);
    TSASR_CHECK_ARG(!(epi_mode == 2 && dbias) || (workspace && workspace_bytes >= tsasr_gemm_bf16_fused_workspace_bytes(M, N)), "tsasr_gemm_bf16_fused: workspace too small");
    GemmPlan pl = plan(M, N, K, 0);
    hipStream_t st = (hipStream_t)stream;
    static const bool log_small = false;
    if (log_small && (long long)cdiv(M, tile_bm(pl.tile)) * cdiv(N, pl.tile == 0 ? 128 : 64) <= 16)
        fprintf(stderr, "[tsasr_gemm_bf16_fused] few tiles: M=%d N=%d K=%d tA=%d tB=%d mode=%d tile=%d\n", M, N, K, transA, transB, epi_mode, pl.tile);
    if (!transA && !transB && g_force_tile < 0) {
        const int bm = tsasr_gemm_big_bm(M, N, K);
        float *colpart = (epi_mode == 2 && dbias) ? (float *)workspace : nullptr;
        if (bm && tsasr_gemm_big_launch(A, B, C, M, N, K, lda, ldb, ldc, epi_mode, bias, y, ldy, slope, p, seed, seed_dev, colpart, mask, st) == 0) {
            if (colpart) tsasr_reduce_submit(colpart, dbias, N, cdiv(M, bm), N, 0, st);
            TSASR_CHECK_LAUNCH("tsasr_gemm_bf16_fused");
            return 0;
        }
    }
    EpiArgs ep{};
    ep.mode = epi_mode; ep.bias = bias; ep.y = (const bf16_t *)y; ep.ldy = ldy; ep.slope = slope; ep.p = p; ep.seed = seed; ep.seed_dev = seed_dev;
    ep.colpart = (epi_mode == 2 && dbias) ? (float *)workspace : nullptr;
    launch_tile<0>(pl.tile, transA, transB, A, B, C, M, N, K, lda, ldb, ldc, 1, pl.kchunk, 0, st, ep);
    if (ep.colpart) tsasr_reduce_submit(ep.colpart, dbias, N, cdiv(M, tile_bm(pl.tile)), N, 0, st);
    TSASR_CHECK_LAUNCH("tsasr_gemm_bf16_fused");
    return 0;
}

}  // extern "C"
