// Grouped weight-gradient GEMM for gfx950: every dW += dy^T . x of a stretch of backward in ONE launch, no split-K.
//
// Replaces (reference: torch autograd's AccumulateGrad over the Linear / Conv1d(k=1) weights of the Conformer block,
// SB/nnet/attention.py:549-553,820-836, SB/lobes/models/transformer/Conformer.py:76-98, SB/nnet/linear.py:64-78; the DDP
// reducer then all-reduces them per bucket, SB/core.py:1464-1484) the per-layer weight-gradient launches of round 1:
// 64x64 output tiles split along the token dimension into fp32 slabs (768 workgroups per GEMM to fill the chip) that a second
// kernel summed - 2.35 GB of slab reads per step, 16 KB of L2->LDS traffic per 64^3 MACs.
// A weight gradient feeds nothing downstream in backward, so the host only QUEUES it (operands kept alive) and flushes a whole
// bucket of the gradient arena at once: a few hundred 256x256 output tiles from dozens of GEMMs fill the 256 CUs with
// no split along the long inner dimension (tokens, 4000-8000), every tile owns its piece of the arena (plain read-add-store:
// deterministic, no slabs, no atomics), and a 256x256x64 k-tile moves 64 KB L2->LDS per 8.4 MFLOP (4x less per FLOP).
//
//   job:  C[M,N] (fp32, ldc) += A[K,M]^T . B[K,N]      A = dy [tokens, out_features], B = x [tokens, in_features], bf16 row-major
//
// Kernel: 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 blocks of v_mfma_f32_32x32x16_bf16 (128 accumulator
// registers); both operands are k-major, so a k-tile is [64 tokens][256 features] (512-byte rows) filled by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, XOR swizzle on the SOURCE chunk, guide 5.4 rule 21) into a 2-slot ring
// (2 x 64 KiB) and consumed through ds_read_b64_tr_b16 (guide T10); one raw s_barrier per k-tile, the next tile's DMA in flight
// behind the 32 MFMAs per wave. Ragged inner dimension (K % 64 != 0, the speaker branch's 4000 tokens): the last tile's rows are
// clamped at the source and zeroed in LDS. Workgroup -> tile: XCD-contiguous ids (guide T1), then a 64-ary search in the job table.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

#define WG_BM 256
#define WG_BN 256
#define WG_THREADS 512
#define WG_ROW_BYTES 512                       // one k-row of a tile: 256 features x 2 B

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_w;

struct WgradJob {
    const bf16_t *A;   // [K, M] row-major (lda)
    const bf16_t *B;   // [K, N] row-major (ldb)
    float *C;          // [M, N] fp32 (ldc), accumulated
    long long lda, ldb, ldc;
    int M, N, K, tile0, tiles_n, pad_;
};

// chunk swizzle of a k-major tile with 512-byte rows: the 4 consecutive k-rows one transposing read touches land in 4 different
// 64-byte bank groups
__device__ __forceinline__ int wg_swz(int row, int ch) { return ch ^ ((row & 3) << 2); }

// One LDS-DMA piece (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds_dst`) as inline asm.
// Through the builtin, hipcc (ROCm 7.2) treats the DMA as a pending LDS store that may alias every later ds_read_b64_tr_b16 and
// drains it (s_waitcnt vmcnt(0)) in front of the first fragment read of the SAME k-tile: transfer and MFMAs ran one after the other
// (stamped loop: 5.4k cycles per k-tile = 4.0k DMA-only + 2.5k MFMA-only, minus almost nothing; round 1's weight-gradient kernel had
// the same wait). Hidden in asm, the only waits are the counted ones below (guide 5.7 item 1: M0 written in the statement that uses it).
__device__ __forceinline__ void wg_dma16(const bf16_t *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// source pointers of this wave's pieces of k-tile 0 of one operand (BK rows of 512 bytes; piece p = LDS rows 2p, 2p+1)
template <int BK>
__device__ __forceinline__ void wg_src(const bf16_t *__restrict__ src, long long ld, int col0, int ncols, int wave, int lane,
                                       const bf16_t *(&g)[BK / 16]) {
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
        const int byte = (i * 8 + wave) * 1024 + lane * 16;
        const int lrow = byte / WG_ROW_BYTES, pos = (byte % WG_ROW_BYTES) / 16;
        g[i] = src + (long long)lrow * ld + min(col0 + wg_swz(lrow, pos) * 8, ncols - 8);
    }
}

// issue pieces [I0, I1) of the k-tile starting at row k0 into the tile at LDS byte address `tile`; clamp_k: rows >= K re-read row K-1
template <int BK, int I0, int I1>
__device__ __forceinline__ void wg_issue(const bf16_t *const (&g)[BK / 16], long long ld, int k0, int K, unsigned tile, int wave, int lane,
                                         bool clamp_k) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        long long koff = (long long)k0 * ld;
        if (clamp_k) {
            const int lrow = ((i * 8 + wave) * 1024 + lane * 16) / WG_ROW_BYTES;
            koff = (long long)(min(k0 + lrow, K - 1) - lrow) * ld;
        }
        wg_dma16(g[i] + koff, __builtin_amdgcn_readfirstlane(tile + (unsigned)(i * 8 + wave) * 1024u));
    }
}

__device__ __forceinline__ bf16x8 wg_frag(const char *tile, int off) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_w *)(tile + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_w *)(tile + off + 4 * WG_ROW_BYTES));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

__device__ unsigned long long g_wg_stamp[4];   // timing experiments (DBG builds): shader cycles / 100 MHz ticks around one workgroup's main loop

// BK rows per ring slot, NST slots: NST - 1 k-tiles are in flight behind the MFMAs of the current one (the operands of a step's
// weight gradients - ~3 GB - come from HBM, not from L2: bytes in flight per CU x 256 CUs / latency is the rate).
template <int BK, int NST, int DBG = 0, bool SPREAD = false>   // DBG (timing experiments only): 1 = DMA and waits only, 2 = fragment reads + MFMAs only, 3 = everything, stamped
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_group_kernel(const WgradJob *__restrict__ jobs, int njobs, int total_tiles) {
    constexpr int TILE_BYTES = BK * WG_ROW_BYTES, SLOT_BYTES = 2 * TILE_BYTES, LPT = 2 * (BK / 16);   // LPT: DMA instructions per wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
    // A launch with fewer workgroups than tiles walks the tiles persistently (vb, vb + gridDim.x, ...): the early launch that runs beside
    // another stream's kernels occupies only that many CUs (a workgroup holds 64-128 KB of LDS: nothing else fits beside it on a CU).
    for (int vb = blockIdx.x; vb < total_tiles; vb += gridDim.x) {
    // XCD-contiguous tile ids: consecutive tiles (same job, same row panel) share one XCD's L2
    int id;
    {
        const int bid = vb, xcd = bid & 7, q = total_tiles >> 3, rem = total_tiles & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    int lo = 0, n = njobs;
    while (n > 1) {   // last job whose first tile <= id: 64-ary search, one lane per probe
        const int step = (n + 63) >> 6, idx = lo + lane * step;
        const bool le = lane * step < n && jobs[idx].tile0 <= id;
        const int seg = __popcll(__ballot(le)) - 1;
        lo += seg * step;
        n = min(step, n - seg * step);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WgradJob jb = jobs[lo];
    const int lt = id - jb.tile0, tn = lt % jb.tiles_n, tm = lt / jb.tiles_n;
    const int m0 = tm * WG_BM, n0 = tn * WG_BN, K = jb.K;
    const int nk = (K + BK - 1) / BK, ktail = K % BK;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    // fragment byte offsets inside a tile for k16-step 0 (step s adds 16 rows = s * 8192 bytes; the second read 4 rows = 2048)
    const int hh = lane >> 5, mhalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3, r = lane & 31;
    int a_off[4], b_off[2];
    {
        const int row = 8 * hh + q4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wm * 128 + 32 * i + 16 * mhalf + 4 * p4;
            a_off[i] = row * WG_ROW_BYTES + wg_swz(row, col >> 3) * 16 + (col & 7) * 2;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + 32 * j + 16 * mhalf + 4 * p4;
            b_off[j] = row * WG_ROW_BYTES + wg_swz(row, col >> 3) * 16 + (col & 7) * 2;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const bf16_t *ga[BK / 16], *gb[BK / 16];
    wg_src<BK>(jb.A, jb.lda, m0, jb.M, wave, lane, ga);
    wg_src<BK>(jb.B, jb.ldb, n0, jb.N, wave, lane, gb);
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk && DBG != 2) {
            const bool last = (t + 1 == nk) && ktail;
            wg_issue<BK, 0, BK / 16>(ga, jb.lda, t * BK, K, lds0 + t * SLOT_BYTES, wave, lane, last);
            wg_issue<BK, 0, BK / 16>(gb, jb.ldb, t * BK, K, lds0 + t * SLOT_BYTES + TILE_BYTES, wave, lane, last);
        }
    unsigned long long t0c = 0, t0r = 0;
    if (DBG != 0) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed once only the newer tiles' DMA instructions are outstanding
        const int newer = min(nk - 1 - kt, NST - 2);
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // everyone's pieces landed; everyone is done reading slot (kt - 1) % NST
        char *cur = smem + (kt % NST) * SLOT_BYTES;
        asm volatile("" ::: "memory");
        // the next free slot's DMA: ISSUE_SPREAD = one piece per operand in front of each k16 step's MFMAs, else one burst here
        const bool more = kt + NST - 1 < nk && DBG != 2;
        const unsigned nxt = lds0 + ((kt + NST - 1) % NST) * SLOT_BYTES;
        const bool nlast = (kt + NST == nk) && ktail;
        const int nk0 = (kt + NST - 1) * BK;
        if (more && !SPREAD) {
            wg_issue<BK, 0, BK / 16>(ga, jb.lda, nk0, K, nxt, wave, lane, nlast);
            wg_issue<BK, 0, BK / 16>(gb, jb.ldb, nk0, K, nxt + TILE_BYTES, wave, lane, nlast);
        }
        if (kt + 1 == nk && ktail) {   // rows beyond K hold clamped copies of row K-1: zero them in both operands
            for (int c = threadIdx.x; c < (BK - ktail) * (WG_ROW_BYTES / 16) * 2; c += WG_THREADS) {
                const int op = c / ((BK - ktail) * (WG_ROW_BYTES / 16)), cc = c % ((BK - ktail) * (WG_ROW_BYTES / 16));
                *reinterpret_cast<uint4 *>(cur + op * TILE_BYTES + ktail * WG_ROW_BYTES + cc * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
        const char *as = cur, *bs = cur + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            if (more && SPREAD) {
                if (s == 0) { wg_issue<BK, 0, 1>(ga, jb.lda, nk0, K, nxt, wave, lane, nlast); wg_issue<BK, 0, 1>(gb, jb.ldb, nk0, K, nxt + TILE_BYTES, wave, lane, nlast); }
                if (s == 1) { wg_issue<BK, 1, 2>(ga, jb.lda, nk0, K, nxt, wave, lane, nlast); wg_issue<BK, 1, 2>(gb, jb.ldb, nk0, K, nxt + TILE_BYTES, wave, lane, nlast); }
                if constexpr (BK / 16 > 2) {
                    if (s == 2) { wg_issue<BK, 2, 3>(ga, jb.lda, nk0, K, nxt, wave, lane, nlast); wg_issue<BK, 2, 3>(gb, jb.ldb, nk0, K, nxt + TILE_BYTES, wave, lane, nlast); }
                    if (s == 3) { wg_issue<BK, 3, 4>(ga, jb.lda, nk0, K, nxt, wave, lane, nlast); wg_issue<BK, 3, 4>(gb, jb.ldb, nk0, K, nxt + TILE_BYTES, wave, lane, nlast); }
                }
            }
            if (DBG == 1) continue;
            bf16x8 af[4], bfr[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = wg_frag(as, a_off[i] + s * 16 * WG_ROW_BYTES);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = wg_frag(bs, b_off[j] + s * 16 * WG_ROW_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    if (DBG != 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        g_wg_stamp[0] = __builtin_amdgcn_s_memtime() - t0c;
        g_wg_stamp[1] = __builtin_amdgcn_s_memrealtime() - t0r;
        g_wg_stamp[2] = nk;
    }
    // C += acc: accumulator layout = lane owns one column n, 16 rows per block; a wave-instruction touches 2 rows x 128 bytes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = n0 + wn * 64 + 32 * j + r;
        if (nn >= jb.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float old[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int mm = min(m0 + wm * 128 + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh, jb.M - 1);
                old[g] = jb.C[(long long)mm * jb.ldc + nn];
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int mm = m0 + wm * 128 + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh;
                if (mm < jb.M) jb.C[(long long)mm * jb.ldc + nn] = old[g] + acc[i][j][g];
            }
        }
    }
    if (vb + (int)gridDim.x < total_tiles) {     // another tile follows: its DMA reuses the slots and counts on an empty vmcnt
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    }   // tile loop
}

static std::vector<WgradJob> g_wjobs;
static int g_next_slots = 0, g_next_wgs = 0;

extern "C" {

/* Queue dW[M,N] (fp32, row stride ldc) += dy[K,M]^T . x[K,N] (bf16 row-major, row strides ld_dy / ld_x; K = tokens). Nothing is
 * launched: dy, x and dW must stay alive and unmodified until tsasr_wgrad_flush. M, N, ld_dy, ld_x multiples of 8, M, N >= 8,
 * pointers 16-byte aligned. (Replaces the per-weight AccumulateGrad of the reference; see the file header.) */
int tsasr_wgrad_queue(const void *dy, const void *x, float *dW, int M, int N, int K, long long ld_dy, long long ld_x, long long ldc) {
    TSASR_CHECK_ARG(dy && x && dW, "tsasr_wgrad_queue: null pointer");
    TSASR_CHECK_ARG(M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && ld_dy % 8 == 0 && ld_x % 8 == 0,
                    "tsasr_wgrad_queue: M, N and the row strides must be multiples of 8 (M=%d N=%d K=%d ld_dy=%lld ld_x=%lld)", M, N, K, ld_dy, ld_x);
    TSASR_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dW & 3) == 0, "tsasr_wgrad_queue: misaligned pointer");
    WgradJob j{(const bf16_t *)dy, (const bf16_t *)x, dW, ld_dy, ld_x, ldc, M, N, K, 0, cdiv(N, WG_BN), 0};
    g_wjobs.push_back(j);
    return 0;
}

int tsasr_wgrad_pending(void) { return (int)g_wjobs.size(); }

size_t tsasr_wgrad_table_bytes(int max_jobs) { return (size_t)max_jobs * sizeof(WgradJob); }

/* Run every queued weight gradient in ONE launch on `stream` (which must be ordered after the producers of every operand).
 * table_host: PINNED host memory, table_dev: device memory, both >= tsasr_wgrad_table_bytes(tsasr_wgrad_pending()); the table is
 * copied host -> device on `stream`, except while `stream` is being captured: then only table_host is filled and the caller
 * uploads it after the capture (both must outlive the graph; same protocol as tsasr_reduce_flush). */
int tsasr_wgrad_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream) {
    if (g_wjobs.empty()) return 0;
    hipStream_t st = (hipStream_t)stream;
    // longest inner dimension first (stable): the long tiles start first, the short ones fill the tail of the launch
    std::stable_sort(g_wjobs.begin(), g_wjobs.end(), [](const WgradJob &a, const WgradJob &b) { return a.K > b.K; });
    int tiles = 0;
    for (auto &j : g_wjobs) {
        j.tile0 = tiles;
        tiles += cdiv(j.M, WG_BM) * j.tiles_n;
    }
    const size_t need = g_wjobs.size() * sizeof(WgradJob);
    TSASR_CHECK_ARG(table_host && table_dev && table_bytes >= need, "tsasr_wgrad_flush: job table too small (%zu < %zu bytes)", table_bytes, need);
    memcpy(table_host, g_wjobs.data(), need);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap == hipStreamCaptureStatusNone) {
        hipError_t e = hipMemcpyAsync(table_dev, table_host, need, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            tsasr_set_error("tsasr_wgrad_flush: job table upload failed: %s", hipGetErrorString(e));
            return TSASR_E_LAUNCH;
        }
    }
    static const int variant = 1;   // bit 0: 32-row slots x 4 instead of 64-row x 2; bit 1: spread DMA issue (A/B)
    static const int dbg = getenv("TSASR_WGRAD_DEBUG") ? atoi(getenv("TSASR_WGRAD_DEBUG")) : 0;      // timing experiments: see DBG
    void (*kern)(const WgradJob *, int, int) = nullptr;
#define WG_PICK(BK_, NST_) \
    (dbg == 1 ? wgrad_group_kernel<BK_, NST_, 1, false> : dbg == 2 ? wgrad_group_kernel<BK_, NST_, 2, false> : \
     dbg == 3 ? ((variant & 2) ? wgrad_group_kernel<BK_, NST_, 3, true> : wgrad_group_kernel<BK_, NST_, 3, false>) : \
                ((variant & 2) ? wgrad_group_kernel<BK_, NST_, 0, true> : wgrad_group_kernel<BK_, NST_, 0, false>))
    kern = (variant & 1) ? WG_PICK(32, 4) : WG_PICK(64, 2);
    int lds_bytes = 128 * 1024;
    if (g_next_slots == 2 && dbg == 0) {    // a launch meant to run BESIDE other kernels: 64 KB of LDS (two 32-row slots) leaves room on the CU
        kern = wgrad_group_kernel<32, 2, 0, false>;
        lds_bytes = 64 * 1024;
    }
    g_next_slots = 0;
    const int wgs = (g_next_wgs > 0 && g_next_wgs < tiles && dbg == 0) ? g_next_wgs : tiles;
    g_next_wgs = 0;
#undef WG_PICK
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    kern<<<wgs, WG_THREADS, lds_bytes, st>>>((const WgradJob *)table_dev, (int)g_wjobs.size(), tiles);
    g_wjobs.clear();
    TSASR_CHECK_LAUNCH("tsasr_wgrad_flush");
    return 0;
}

/* Timing experiments: {shader cycles, 100 MHz ticks, k-tiles} of workgroup 0's main loop in the last TSASR_WGRAD_DEBUG launch. */
int tsasr_wgrad_debug_read(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg_stamp), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : TSASR_E_LAUNCH;
}

/* The NEXT tsasr_wgrad_flush uses `slots` LDS-DMA slots of 32 rows (2 = 64 KB of LDS per workgroup instead of 128 KB: slower alone, but
 * other kernels' workgroups fit beside it on a CU); 0 = default. One-shot. */
void tsasr_wgrad_next_flush_slots(int slots) { g_next_slots = slots; }

/* The NEXT tsasr_wgrad_flush launches at most `wgs` workgroups, which walk the tiles persistently (0 = one workgroup per tile):
 * a launch that should leave the other CUs to another stream's kernels. One-shot. */
void tsasr_wgrad_next_flush_wgs(int wgs) { g_next_wgs = wgs; }

/* Drop the queue without running it (error paths / tests). */
void tsasr_wgrad_discard(void) { g_wjobs.clear(); }

}  // extern "C"
