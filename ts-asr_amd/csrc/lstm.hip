// LSTM cell kernels for the transducer's prediction network on gfx950 (one launch per time step and direction; the
// recurrent matrix product runs on csrc/gemm.hip - every launch is hipGraph-capturable, unlike MIOpen's RNN path).
//
// Replaces torch.nn.LSTM as wrapped by speechbrain/nnet/RNN.py:244-278 (single layer, batch_first, gate order i,f,g,o).
//   gates_t = x_t W_ih^T + b_ih + b_hh + h_{t-1} W_hh^T ;  c_t = sig(f) c_{t-1} + sig(i) tanh(g) ;  h_t = sig(o) tanh(c_t)
// Layout: gates [B, U, H, 4] fp32 - the four gates (i,f,g,o) of a unit are ONE float4 (the caller permutes the rows of W_ih
// accordingly; pre-activations in, ACTIVATED gates out - kept for the backward); dgates [B, U, 4H] gate-major; c [B, U, H] fp32,
// h [B, U, H] in the activation dtype (it is the next step's GEMM operand and the layer output).
#include <algorithm>

#include <stdlib.h>

#include "common.h"

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence: 5 transcendentals per cell sit on the recurrence's critical path
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { const float e = __expf(-2.f * fabsf(x)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); return x < 0.f ? -t : t; }

template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float *__restrict__ gates, float *__restrict__ c, T *__restrict__ h, int B,
                                                            int U, int H, int t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    float4 *g = reinterpret_cast<float4 *>(gates + (((long long)b * U + t) * H + k) * 4);
    const float4 pre4 = *g;
    const float gi = sigm(pre4.x), gf = sigm(pre4.y), gg = tanh_fast(pre4.z), go = sigm(pre4.w);
    const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
    const float cn = gf * cp + gi * gg;
    *g = make_float4(gi, gf, gg, go);
    c[((long long)b * U + t) * H + k] = cn;
    st1(h + ((long long)b * U + t) * H + k, go * tanh_fast(cn));
}

// dgates_t (pre-activation gradients, activation dtype: operand of the dh and dW GEMMs), dc carried in dc_io [B,H] fp32
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ c,
                                                            const T *__restrict__ dout, const float *__restrict__ dh_rec,
                                                            float *__restrict__ dc_io, T *__restrict__ dgates, int B, int U, int H, int t,
                                                            int last) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    const float4 g4 = *reinterpret_cast<const float4 *>(gates + (((long long)b * U + t) * H + k) * 4);
    const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
    const float cn = c[((long long)b * U + t) * H + k];
    const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
    const float tc = tanh_fast(cn);
    const float dh = ld1(dout + ((long long)b * U + t) * H + k) + (last ? 0.f : dh_rec[i]);
    const float dc = dh * go * (1.f - tc * tc) + (last ? 0.f : dc_io[i]);
    T *dg = dgates + ((long long)b * U + t) * 4 * H;
    st1(dg + k, dc * gg * gi * (1.f - gi));
    st1(dg + H + k, dc * cp * gf * (1.f - gf));
    st1(dg + 2 * H + k, dc * gi * (1.f - gg * gg));
    st1(dg + 3 * H + k, dh * tc * go * (1.f - go));
    dc_io[i] = dc * gf;
}


// =====================================================================================================================
// Fused time steps: recurrent product + cell update in ONE launch per step and direction.
//   workgroup = 16 hidden units (= 64 gate columns i,f,g,o of those units) x up to 32 batch rows; the K range of the product
//   is split over the 4 waves (operand fragments straight from global/L2: every element is used once per workgroup), partial
//   accumulators are combined through LDS, then each (row, unit) pair does its cell math. 121 steps -> 121 launches per
//   direction (hipGraph nodes) instead of 242 + slab reductions.
// =====================================================================================================================
#define LS_UN 16   // hidden units per workgroup

template <typename T>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float *__restrict__ gates, float *__restrict__ c, T *__restrict__ h,
                                                            const bf16_t *__restrict__ whh /*[4H,H]*/, int B, int U, int H, int t) {
    __shared__ float red[4][32][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int u0 = blockIdx.x * LS_UN, b0 = blockIdx.y * 32;
    f32x16 acc[2];
    acc[0] = acc[1] = (f32x16){0};
    if (t > 0) {
        const int ks = H / 16, per = (ks + 3) / 4;
        const int brow = min(b0 + r, B - 1);
        const T *hrow = h + ((long long)brow * U + (t - 1)) * H;
        const int s_end = min(ks, (wave + 1) * per);
        for (int sb = wave * per; sb < s_end; sb += 8) {   // issue 8 k-steps of operand loads, then their 16 MFMAs (one L2 latency per chunk)
            bf16x8 af[8], bf0[8], bf1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = min(sb + q, s_end - 1);
                float a8[8];
                ld8(hrow + 16 * s + 8 * hh, a8);
#pragma unroll
                for (int j = 0; j < 8; ++j) af[q][j] = (bf16_t)a8[j];
                bf0[q] = *reinterpret_cast<const bf16x8 *>(whh + ((long long)(r >> 4) * H + u0 + (r & 15)) * H + 16 * s + 8 * hh);
                bf1[q] = *reinterpret_cast<const bf16x8 *>(whh + ((long long)(2 + (r >> 4)) * H + u0 + (r & 15)) * H + 16 * s + 8 * hh);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (sb + q < s_end) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bf0[q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bf1[q], acc[1], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 16; ++g) red[wave][(g & 3) + 8 * (g >> 2) + 4 * hh][32 * cb + r] = acc[cb][g];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * LS_UN; i += 256) {
        const int bl = i / LS_UN, j = i % LS_UN, b = b0 + bl, k = u0 + j;
        if (b >= B) continue;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = red[0][bl][16 * g + j] + red[1][bl][16 * g + j] + red[2][bl][16 * g + j] + red[3][bl][16 * g + j];
        float4 *gp = reinterpret_cast<float4 *>(gates + (((long long)b * U + t) * H + k) * 4);
        const float4 gx4 = *gp;
        const float gi = sigm(gx4.x + pre[0]), gf = sigm(gx4.y + pre[1]), gg = tanh_fast(gx4.z + pre[2]), go = sigm(gx4.w + pre[3]);
        const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
        const float cn = gf * cp + gi * gg;
        *gp = make_float4(gi, gf, gg, go);
        c[((long long)b * U + t) * H + k] = cn;
        st1(h + ((long long)b * U + t) * H + k, go * tanh_fast(cn));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ c,
                                                            const T *__restrict__ dout, T *__restrict__ dgates,
                                                            const bf16_t *__restrict__ whhT /*[H,4H]*/, float *__restrict__ dc_io, int B,
                                                            int U, int H, int t) {
    __shared__ float red[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int u0 = blockIdx.x * LS_UN, b0 = blockIdx.y * 32;
    const bool last = (t == U - 1);
    f32x16 acc = {0};
    if (!last) {   // dh_rec[b, u0+j] = sum_c dgates[b, t+1, c] * W_hh[c, u0+j]
        const int K4 = 4 * H, ks = K4 / 16, per = (ks + 3) / 4;
        const int brow = min(b0 + r, B - 1);
        const T *drow = dgates + ((long long)brow * U + (t + 1)) * K4;
        const bf16_t *wrow = whhT + (long long)(u0 + (r & 15)) * K4;
        const int s_end = min(ks, (wave + 1) * per);
        for (int sb = wave * per; sb < s_end; sb += 8) {
            bf16x8 af[8], bfr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = min(sb + q, s_end - 1);
                float a8[8];
                ld8(drow + 16 * s + 8 * hh, a8);
#pragma unroll
                for (int j = 0; j < 8; ++j) af[q][j] = (bf16_t)a8[j];
                bfr[q] = *reinterpret_cast<const bf16x8 *>(wrow + 16 * s + 8 * hh);   // columns 16..31 duplicate 0..15 (unused)
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (sb + q < s_end) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bfr[q], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) red[wave][(g & 3) + 8 * (g >> 2) + 4 * hh][r] = acc[g];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * LS_UN; i += 256) {
        const int bl = i / LS_UN, j = i % LS_UN, b = b0 + bl, k = u0 + j;
        if (b >= B) continue;
        const float dh_rec = red[0][bl][j] + red[1][bl][j] + red[2][bl][j] + red[3][bl][j];
        const float4 g4 = *reinterpret_cast<const float4 *>(gates + (((long long)b * U + t) * H + k) * 4);
        const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
        const float cn = c[((long long)b * U + t) * H + k];
        const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
        const float tc = tanh_fast(cn);
        const float dh = ld1(dout + ((long long)b * U + t) * H + k) + (last ? 0.f : dh_rec);
        const float dc = dh * go * (1.f - tc * tc) + (last ? 0.f : dc_io[(long long)b * H + k]);
        T *dg = dgates + ((long long)b * U + t) * 4 * H;
        st1(dg + k, dc * gg * gi * (1.f - gi));
        st1(dg + H + k, dc * cp * gf * (1.f - gf));
        st1(dg + 2 * H + k, dc * gi * (1.f - gg * gg));
        st1(dg + 3 * H + k, dh * tc * go * (1.f - go));
        dc_io[(long long)b * H + k] = dc * gf;
    }
}

// =====================================================================================================================
// Whole-sequence persistent kernels: ONE launch per direction instead of one per time step.
//   The per-step launches above cost 11-12 us each (121 steps x 2 directions = 2.9 ms of a 31 ms training step) although a
//   step is 67 MFLOP: every launch re-reads its W_hh slice from L2 and pays a kernel boundary. Here a workgroup owns 32 hidden
//   units for the whole sequence, keeps its slice of W_hh (forward: [4 x 32, H]; backward: W_hh^T [32, 4H], K split over the
//   four waves) in REGISTERS as MFMA A-fragments, and the workgroups of one 32-row batch group exchange h_t (forward) / the gate
//   gradients dG_t (backward) once per step through global memory:
//     producer: payload tile -> LDS -> 16-byte write-through (sc1) stores, each wave-instruction writing 1 KiB of whole
//               lines -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane's agent-scope atomic add;
//     consumer: ONE lane polls the counter with sc1 loads (s_sleep between polls, bounded) -> workgroup barrier -> every
//               load of the payload is an sc1 16-byte buffer load straight to registers (MFMA B-fragments).
//   (gfx950's eight XCD-private L2s are not coherent with each other: plain loads of another workgroup's stores are stale.)
//   Each step's payload has its own location (no reuse inside a launch); the counter is zeroed by a tiny fill kernel ahead of the
//   launch. Grid = (H/32, ceil(B/16)) <= the CU count, 1 workgroup per CU (96 KB of dynamic LDS keeps a second one away), so
//   every workgroup is resident and the per-step waits cannot deadlock; a poll that never matches gives up after ~1 s,
//   raises the error word and poisons the output with NaN (the step's finite-check then rejects the update).
// =====================================================================================================================
#define LQ_UN 32
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool lq_wait(unsigned *ctr, unsigned target, unsigned *err) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 23)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// BR = batch rows per exchange group (a power of two <= 32). The recurrence is independent per batch row, so a launch runs
// ceil(B / BR) groups of H/32 workgroups side by side; a smaller BR shrinks what a workgroup pulls through the exchange per step
// and its cell work, but more workgroups poll and arrive per step: measured per-step cycles of the forward kernel (H = 512,
// B = 32) 8.8k / 8.7k / 10.3k at BR = 32 / 16 / 8 (the wait for h_{t-1} grows 2.0k -> 2.7k -> 5.8k), backward 809 / 689 / 797 us.
template <int H, int BR>
__global__ __launch_bounds__(256, 1) void lstm_seq_fwd_kernel(float *__restrict__ gates, float *__restrict__ c, bf16_t *__restrict__ h,
                                                              const bf16_t *__restrict__ whh /*[4H,H]*/, bf16_t *xh /*[G][U][H/32][BR][32]*/,
                                                              unsigned *sync, int B, int U, int xcd) {
    // xcd: the launch has 8 x the workgroups and group bg keeps those dealt to ONE XCD (workgroup n of a dispatch goes to XCD n mod 8): the
    // exchange then stays inside one L2 instead of crossing the fabric. The other seven eighths leave at once.
    if (xcd && (int)(blockIdx.x & 7) != (int)(blockIdx.y & 7)) return;
    constexpr int NWG = H / LQ_UN, PP = BR * 32 / 256, TILE = BR * 32;
    static_assert(PP >= 1 && (BR & (BR - 1)) == 0 && BR <= 32, "BR in {8, 16, 32}");
    __shared__ __attribute__((aligned(16))) float pre[BR][LQ_UN * 4 + 4];     // [batch][unit*4 + gate]
    __shared__ __attribute__((aligned(16))) bf16_t hs[BR][LQ_UN];              // this step's h tile, laid out as it is published
    __shared__ __attribute__((aligned(16))) bf16_t hl[(H / 32) * BR * 40];     // h_{t-1}: [unit block][batch row][32 + 8 pad]
    __shared__ int ok_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd ? blockIdx.x >> 3 : blockIdx.x, bg = blockIdx.y, u0 = wg * LQ_UN, b0 = bg * BR;
    unsigned *ctr = sync + 2 * bg, *err = sync + 2 * bg + 1;
    bf16_t *xh_g = xh + (size_t)bg * U * H * BR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xh_g, 0, U * H * BR * 2, 0x00020000);
    // A operand (W_hh rows): MFMA row m of row block rb = gate (m & 3) of unit u0 + 8*wave + 4*rb + (m >> 2) -> the four gates of a
    // unit land in the four accumulator registers of ONE lane
    // v_mfma_f32_16x16x32_bf16: the batch side of the product has at most 16 rows, so the 32-column form spent half of every MFMA and
    // of every B-fragment read on repeated columns. Two row blocks of 16 (4 units x 4 gates each) share one B fragment per 32-wide k step.
    static_assert(BR <= 16, "the 16-column MFMA form covers groups of at most 16 batch rows");
    constexpr int KS2 = H / 32;
    const int m16 = lane & 15, kg = lane >> 4;
    bf16x8 af[2][KS2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const bf16_t *wrow = whh + ((long long)(m16 & 3) * H + u0 + 8 * wave + 4 * rb + (m16 >> 2)) * H;
#pragma unroll
        for (int s = 0; s < KS2; ++s) af[rb][s] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * s + 8 * kg);
    }
    float cprev[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) cprev[j] = 0.f;
    bool failed = false;
#ifdef LQ_PROFILE
    long long acc_t[5] = {0, 0, 0, 0, 0}, t_prev = clock64();
#define LQ_STAMP(i) do { const long long n_ = clock64(); acc_t[i] += n_ - t_prev; t_prev = n_; } while (0)
#else
#define LQ_STAMP(i)
#endif
#pragma unroll 1
    for (int t = 0; t < U; ++t) {
        float gx[PP][4];
#pragma unroll
        for (int j = 0; j < PP; ++j) {   // x-part pre-activations of this thread's (batch, unit) pairs: in flight across the wait
            const int p = tid + 256 * j, bl = p >> 5, ul = p & 31;
            const float4 g4 = *reinterpret_cast<const float4 *>(gates + (((long long)min(b0 + bl, B - 1) * U + t) * H + u0 + ul) * 4);
            gx[j][0] = g4.x; gx[j][1] = g4.y; gx[j][2] = g4.z; gx[j][3] = g4.w;
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (t > 0) {
            if (tid == 0) ok_flag = lq_wait(ctr, (unsigned)(NWG * t), err) ? 1 : 0;
            __syncthreads();
            if (!ok_flag) { failed = true; break; }
            LQ_STAMP(0);   // wait for h_{t-1}
            // h_{t-1} [BR x H] enters the workgroup ONCE (each thread NLD coalesced 16-byte sc1 loads, all requested together),
            // is laid out in LDS as [unit block][batch row][32 + 8 pad] and feeds the four waves' B fragments from there
            constexpr int NLD = (H * BR / 8 + 255) / 256;
            u32x4 raw[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((t - 1) * NWG * TILE + min(tid + 256 * i, H * BR / 8 - 1) * 8) * 2, 0, 16);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int ch = tid + 256 * i;
                if (ch < H * BR / 8) *reinterpret_cast<u32x4 *>(&hl[((ch / (BR * 4)) * BR + ((ch % (BR * 4)) >> 2)) * 40 + (ch & 3) * 8]) = raw[i];
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8 *>(&hl[(s * BR + (m16 & (BR - 1))) * 40 + 8 * kg]);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][s], bfrag, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][s], bfrag, acc[1], 0, 0, 0);
            }
        }
        LQ_STAMP(1);   // operand loads + MFMA
        if (m16 < BR) {   // accumulator: column = batch row m16, rows 4*kg .. +3 of block rb = the four gates of unit 8*wave + 4*rb + kg
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                *reinterpret_cast<float4 *>(&pre[m16][(8 * wave + 4 * rb + kg) * 4]) = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
        }
        __syncthreads();
        float4 gact[PP];
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int p = tid + 256 * j, bl = p >> 5, ul = p & 31, b = b0 + bl;
            const float4 pr = *reinterpret_cast<const float4 *>(&pre[bl][ul * 4]);
            const float gi = sigm(gx[j][0] + pr.x), gf = sigm(gx[j][1] + pr.y), gg = tanh_fast(gx[j][2] + pr.z), go = sigm(gx[j][3] + pr.w);
            const float cn = gf * cprev[j] + gi * gg;
            cprev[j] = cn;
            const float hv = go * tanh_fast(cn);
            gact[j] = make_float4(gi, gf, gg, go);     // stored AFTER the arrival below: nobody waits for them, the drain in front of it must not
            hs[bl][ul] = (bf16_t)(b < B ? hv : 0.f);
        }
        __syncthreads();
        LQ_STAMP(2);   // cell math
        u32x4 hv16 = {0u, 0u, 0u, 0u};
        if (tid < BR * 4) {   // BR*64-byte tile: whole 128-byte lines per wave-instruction, write-through
            const int ch = tid, bl = ch >> 2, part = ch & 3;
            hv16 = *reinterpret_cast<const u32x4 *>(&hs[bl][part * 8]);
            __builtin_amdgcn_raw_buffer_store_b128(hv16, rs, ((t * NWG + wg) * TILE + ch * 8) * 2, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // only the exchange tile is in flight here
        __syncthreads();
        LQ_STAMP(3);   // publish + drain
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the step's outputs for the backward / the caller: written while the other workgroups are already being waited for
        if (tid < BR * 4) {
            const int bl = tid >> 2, part = tid & 3;
            if (b0 + bl < B) *reinterpret_cast<u32x4 *>(h + ((long long)(b0 + bl) * U + t) * H + u0 + part * 8) = hv16;
        }
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int p = tid + 256 * j, bl = p >> 5, ul = p & 31, b = b0 + bl;
            if (b < B) {
                *reinterpret_cast<float4 *>(gates + (((long long)b * U + t) * H + u0 + ul) * 4) = gact[j];
                c[((long long)b * U + t) * H + u0 + ul] = cprev[j];
            }
        }
        LQ_STAMP(4);
    }
#ifdef LQ_PROFILE
    if (tid == 0 && wg == 3 && bg == 0) for (int i = 0; i < 5; ++i) reinterpret_cast<long long *>(sync + 32)[i] = acc_t[i];
#endif
    if (failed && tid == 0) h[((long long)min(b0, B - 1) * U + U - 1) * H + u0] = (bf16_t)__builtin_nanf("");
}

template <int H, int BR>
__global__ __launch_bounds__(256, 1) void lstm_seq_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ c,
                                                              const bf16_t *__restrict__ dout, bf16_t *__restrict__ dgates,
                                                              const bf16_t *__restrict__ whhT /*[H,4H]*/, bf16_t *xg /*[G][U][H/32][BR][128]*/,
                                                              unsigned *sync, int B, int U, int xcd) {
    if (xcd && (int)(blockIdx.x & 7) != (int)(blockIdx.y & 7)) return;     // (see the forward kernel)
    constexpr int K4 = 4 * H, NWG = H / LQ_UN, PP = BR * 32 / 256, TILE = BR * 128;
    constexpr int XLD = 136;                                               // staged row: 128 bf16 + 8 pad (16-byte slots rotate per row)
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];         // 96 KB (also what keeps one workgroup per CU)
    bf16_t *xl = reinterpret_cast<bf16_t *>(dyn_lds);                      // dgates_{t+1}: [H/32 workgroups x BR rows][XLD]
    static_assert((H / LQ_UN) * BR * XLD * 2 <= 96 * 1024, "staged exchange tile must fit the dynamic LDS");
    __shared__ float red[4][32][33];                                       // [wave][batch][unit]
    __shared__ __attribute__((aligned(16))) bf16_t dgt[BR][4 * LQ_UN];     // [batch][gate*32 + unit]: the published tile
    __shared__ int ok_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd ? blockIdx.x >> 3 : blockIdx.x, bg = blockIdx.y, u0 = wg * LQ_UN, b0 = bg * BR;
    unsigned *ctr = sync + 2 * bg, *err = sync + 2 * bg + 1;
    bf16_t *xg_g = xg + (size_t)bg * U * K4 * BR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xg_g, 0, U * K4 * BR * 2, 0x00020000);
    // A operand: row m of row block rb = unit u0 + 16*rb + m of W_hh^T; the reduction index runs over the PUBLISHED order c' = wg'*128 + gate*32 + unit'
    static_assert(BR <= 16, "the 16-column MFMA form covers groups of at most 16 batch rows");
    constexpr int KSW2 = K4 / 32 / 4;                    // 32-wide k steps per wave (v_mfma_f32_16x16x32_bf16, see the forward kernel)
    const int m16 = lane & 15, kg = lane >> 4;
    bf16x8 af[2][KSW2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const bf16_t *wrow = whhT + (long long)(u0 + 16 * rb + m16) * K4;
#pragma unroll
        for (int i = 0; i < KSW2; ++i) {
            const int cp = 32 * (KSW2 * wave + i) + 8 * kg;
            af[rb][i] = *reinterpret_cast<const bf16x8 *>(wrow + ((cp & 127) >> 5) * H + (cp >> 7) * 32 + (cp & 31));
        }
    }
    float dcar[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) dcar[j] = 0.f;
    bool failed = false;
#ifdef LQ_PROFILE
    long long bacc_t[6] = {0, 0, 0, 0, 0, 0}, bt_prev = clock64();
#define LQB_STAMP(i) do { const long long n_ = clock64(); bacc_t[i] += n_ - bt_prev; bt_prev = n_; } while (0)
#else
#define LQB_STAMP(i)
#endif
#pragma unroll 1
    for (int t = U - 1; t >= 0; --t) {
        float gv[PP][4], cn[PP], cpv[PP], dov[PP];
#pragma unroll
        for (int j = 0; j < PP; ++j) {   // always-issued loads (clamped), masked below
            const int p = tid + 256 * j, bl = p >> 5, ul = p & 31, b = min(b0 + bl, B - 1);
            const float4 g4 = *reinterpret_cast<const float4 *>(gates + (((long long)b * U + t) * H + u0 + ul) * 4);
            gv[j][0] = g4.x; gv[j][1] = g4.y; gv[j][2] = g4.z; gv[j][3] = g4.w;
            cn[j] = c[((long long)b * U + t) * H + u0 + ul];
            const float cpr = c[((long long)b * U + max(t - 1, 0)) * H + u0 + ul];
            cpv[j] = t > 0 ? cpr : 0.f;
            dov[j] = (float)dout[((long long)b * U + t) * H + u0 + ul];
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const bool last = (t == U - 1);
        if (!last) {
            if (tid == 0) ok_flag = lq_wait(ctr, (unsigned)(NWG * (U - 1 - t)), err) ? 1 : 0;
            __syncthreads();
            if (!ok_flag) { failed = true; break; }
            LQB_STAMP(0);   // input loads issued + wait for dgates_{t+1}
            // dgates_{t+1} [BR x 4H] enters the workgroup ONCE, as coalesced 16-byte loads of distinct addresses, and feeds the waves'
            // B fragments from LDS. (First version: every lane fetched its own fragment - 32 loads per lane whose addresses repeat
            // across lanes and across the 16 workgroups: 128 KB per workgroup and step out of the same 64 KB of L2, 8.8k of the
            // step's 12.6k cycles; the forward kernel has staged h_{t-1} this way since round 1: 3.1k.)
            constexpr int NCH = NWG * BR * 16, NLD = (NCH + 255) / 256;       // 16-byte chunks of one published step
            u32x4 raw[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((t + 1) * NWG * TILE + min(tid + 256 * i, NCH - 1) * 8) * 2, 0, 16);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int ch = tid + 256 * i;                                   // chunk = ((workgroup, batch row), 16-byte piece)
                if (ch < NCH) *reinterpret_cast<u32x4 *>(&xl[(ch >> 4) * XLD + (ch & 15) * 8]) = raw[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KSW2; ++i) {
                const int cp = 32 * (KSW2 * wave + i) + 8 * kg;
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8 *>(&xl[((cp >> 7) * BR + (m16 & (BR - 1))) * XLD + (cp & 127)]);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfrag, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bfrag, acc[1], 0, 0, 0);
            }
        }
        LQB_STAMP(1);   // exchange loads + MFMA (issue)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)   // accumulator: column = batch row m16, rows = units 16*rb + 4*kg .. +3
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave][m16][16 * rb + 4 * kg + j] = acc[rb][j];
        __syncthreads();
        LQB_STAMP(2);   // accumulators to LDS (waits for the MFMAs and their operands) + barrier
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int p = tid + 256 * j, bl = p >> 5, ul = p & 31;
            const float dh_rec = red[0][bl][ul] + red[1][bl][ul] + red[2][bl][ul] + red[3][bl][ul];
            const float gi = gv[j][0], gf = gv[j][1], gg = gv[j][2], go = gv[j][3];
            const float tc = tanh_fast(cn[j]);
            const float dh = dov[j] + (last ? 0.f : dh_rec);
            const float dc = dh * go * (1.f - tc * tc) + dcar[j];
            const bool live = b0 + bl < B;
            dgt[bl][ul] = (bf16_t)(live ? dc * gg * gi * (1.f - gi) : 0.f);
            dgt[bl][LQ_UN + ul] = (bf16_t)(live ? dc * cpv[j] * gf * (1.f - gf) : 0.f);
            dgt[bl][2 * LQ_UN + ul] = (bf16_t)(live ? dc * gi * (1.f - gg * gg) : 0.f);
            dgt[bl][3 * LQ_UN + ul] = (bf16_t)(live ? dh * tc * go * (1.f - go) : 0.f);
            dcar[j] = dc * gf;
        }
        __syncthreads();
        LQB_STAMP(3);   // cell backward (waits for this step's gates / c / dout) + barrier
        constexpr int NST = (BR * 16 + 255) / 256;
        u32x4 dv16[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) {   // BR*256-byte tile: wave-instructions of 1 KiB, write-through
            const int ch = tid + 256 * k;
            dv16[k] = (u32x4){0u, 0u, 0u, 0u};
            if (ch < BR * 16) {
                dv16[k] = *reinterpret_cast<const u32x4 *>(&dgt[ch >> 4][(ch & 15) * 8]);
                __builtin_amdgcn_raw_buffer_store_b128(dv16[k], rs, ((t * NWG + wg) * TILE + ch * 8) * 2, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // only the exchange tile (and this step's input loads) are in flight here
        __syncthreads();
        LQB_STAMP(4);   // publish + drain
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the gate gradients for the input-side GEMMs: nobody in this kernel reads them, so they go out after the arrival
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int ch = tid + 256 * k, bl = ch >> 4, part = ch & 15;
            if (ch < BR * 16 && b0 + bl < B)
                *reinterpret_cast<u32x4 *>(dgates + ((long long)(b0 + bl) * U + t) * K4 + (part >> 2) * H + u0 + (part & 3) * 8) = dv16[k];
        }
    }
#ifdef LQ_PROFILE
    if (tid == 0 && wg == 3 && bg == 0) for (int i = 0; i < 5; ++i) reinterpret_cast<long long *>(sync + 32)[i] = bacc_t[i];
#endif
    if (failed && tid == 0) dgates[((long long)min(b0, B - 1) * U) * K4 + u0] = (bf16_t)__builtin_nanf("");
}

// =====================================================================================================================
// ONE utterance (B = 1: the long-form configuration, U ~ 2000 target tokens).
//   With a single batch row the kernels above spend a step on their rendezvous, not on its 2 MFLOP: per step of the forward kernel
//   (cycles at 2.4 GHz, B = 1, U = 1920) wait for the arrival counter 1920, exchange tile through LDS + MFMAs 1840, cell 545, publish +
//   drain 680, outputs 290 = 2.2 us: a counter round trip, a payload round trip and a store acknowledgement in series, five
//   workgroup barriers. What the exchange itself costs was measured on its skeleton (tools/scratch/xcc/ring_probe.hip: 16 workgroups
//   publish 64 bytes each of a 1 KB row and poll the previous row until complete): 1920 cycles per step with agent-scope (write-through)
//   stores, 920 with plain stores when the workgroups share an XCD - whose L2 then is the meeting point; across XCDs plain stores are
//   never seen. Here (forward 4.2 -> 2.0 ms, backward 5.2 -> 2.7 ms at U = 1920):
//   * 16 workgroups of 6 waves - 4 compute waves owning 8 hidden units each for the whole sequence (W_hh rows in registers as above), one
//     POLLING wave, one LOADING wave - launched as 128 workgroups of which every eighth works: workgroup n of a dispatch runs on XCD
//     n mod 8 (tools/scratch/xcc/xcc_probe.hip), so the sixteen share one L2. That placement is checked, not assumed: every workgroup
//     posts its XCC_ID, and plain exchange stores are used only if all sixteen agree (lq1_same_xcd), agent-scope stores otherwise.
//   * the payload carries its own arrival: h_t (forward) / dG_t (backward) are published straight into the OUTPUT tensors h [U, H] /
//     dgates [U, 4H], which the launch pre-fills with the bf16 pattern 0xFFFF (a NaN no kernel here produces: conversions of finite
//     values never give it, NaNs are quieted to 0x7FC0 | sign). The polling wave reads the row it needs with one agent-scope 16-byte
//     load per lane (backward: four) until no 0xFFFF half is left in it - a piece caught half-written still shows one -, puts it into
//     LDS, and ONE barrier per step hands it to the compute waves as MFMA B fragments. No counter, no drain, no second round trip.
//   * the compute waves issue NO vector-memory loads: on gfx9 a wait for a load also waits for every store issued before it (loads and
//     stores share vmcnt and complete out of order with respect to each other), and the waits the compiler placed at the loop's back edge
//     cost a step 1100 - 1700 cycles. The loading wave fetches the step's operands four steps ahead (first-touch HBM reads) into an LDS
//     ring, the polling wave has nothing but exchange rows in its queue.
//   * every B column carries the same row (forward) / the even and odd columns the two halves of dG_{t+1} (backward: the 16 MFMA rows are
//     8 units x the two halves of the reduction range, dh = D[unit][even] + D[8 + unit][odd], one cross-lane add), so the lanes of a
//     16-lane group hold copies of the sums: each takes ONE unit, and the cell arithmetic is issued once per step instead of once per unit
//     (a vector instruction costs 4 cycles whether 2 or 64 lanes matter).
//   A row that never completes: the polling wave gives up after ~1 s, the workgroup raises the error word and leaves the 0xFFFF (NaN)
//   pattern in what it did not write (the others then time out the same way): the step's finite check rejects the update, ops.lstm
//   raises on the error word.
// =====================================================================================================================
#define LQ1_SPINS (1u << 21)
#define LQ1_WGS 16
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_umax16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ bool has_unwritten(u32x4 v) {   // any of the eight halves still 0xFFFF
    const unsigned m = pk_umax16(pk_umax16(v.x, v.y), pk_umax16(v.z, v.w));
    return (m & 0xFFFFu) == 0xFFFFu || m >= 0xFFFF0000u;
}
__device__ __forceinline__ float lane_value(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

__global__ void fill_words_kernel(unsigned *p, size_t n, unsigned v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

#ifdef LQ_PROFILE
#define LQ1_INIT() long long q_acc[4] = {0, 0, 0, 0}, q_prev = clock64()
#define LQ1_STAMP(i) do { const long long n_ = clock64(); q_acc[i] += n_ - q_prev; q_prev = n_; } while (0)
#define LQ1_DUMP() do { if (tid == 0 && blockIdx.x == 24) for (int i = 0; i < 4; ++i) reinterpret_cast<long long *>(sync + 32)[i] = q_acc[i]; } while (0)
#else
#define LQ1_INIT()
#define LQ1_STAMP(i)
#define LQ1_DUMP()
#endif
// Do all LQ1_WGS working workgroups sit on one XCD (one L2)? Then their exchange stores need no write-through: measured on the exchange skeleton
// (tools/scratch/xcc/ring_probe.hip, 16 workgroups, 1 KB row) 920 cycles per step with plain stores against 1920 with agent-scope stores - and
// plain stores are NOT seen from another XCD. Decided at run time from the hardware's XCC_ID, never assumed: every workgroup posts its id
// (agent scope), reads all of them, and uses plain stores only if they are equal. ids: the EVEN words of the sync block's second half (the odd
// words of the block are error words to the host, include/tsasr_hip.h), zeroed by the launch.
__device__ __forceinline__ bool lq1_same_xcd(unsigned *ids, int w, int *flag_lds) {
    if (threadIdx.x == 0) {
        const unsigned mine = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu) + 1u;
        __hip_atomic_store(ids + 2 * w, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int i = 0; i < LQ1_WGS && same; ++i) {
            unsigned v = 0, spins = 0;
            while ((v = __hip_atomic_load(ids + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
            same = (v == mine);      // (a workgroup that never shows up: agent-scope stores; the row polls report it)
        }
        *flag_lds = same ? 1 : 0;
    }
    __syncthreads();
    return *flag_lds != 0;
}

// one 16-byte piece per lane of a published row, polled until complete; false after LQ1_SPINS rounds
__device__ __forceinline__ bool lq1_poll_row(const __amdgpu_buffer_rsrc_t rs, int byte_off, u32x4 &row) {
    unsigned spins = 0;
    for (;;) {
        asm volatile("" ::: "memory");      // (the load is re-issued every round)
        row = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
        if (__builtin_amdgcn_ballot_w64(has_unwritten(row)) == 0) return true;
        if (++spins > LQ1_SPINS) return false;
    }
}

template <int H>
__global__ __launch_bounds__(384, 1) void lstm_seq1_fwd_kernel(float *__restrict__ gates /*[U][H][4]*/, float *__restrict__ c /*[U][H]*/, bf16_t *h /*[U][H], 0xFFFF-filled*/,
                                                               const bf16_t *__restrict__ whh /*[4H,H]*/, unsigned *sync, int U) {
    static_assert(H == 512 && H == LQ1_WGS * 32, "one 16-byte piece of h_{t-1} per lane; 16 workgroups x 4 waves x 8 units");
    constexpr int KS2 = H / 32;
    if (blockIdx.x & 7) return;
    __shared__ __attribute__((aligned(16))) bf16_t hbuf[2][H];
    __shared__ __attribute__((aligned(16))) float gxr[4][32 * 4];      // x-part pre-activations of the workgroup's 32 units, rows t .. t + 2
    __shared__ int fail_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kg = lane >> 4;
    const int ub = (blockIdx.x >> 3) * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(h, 0, U * H * 2, 0x00020000);
    if (tid == 0) fail_flag = 0;
    __shared__ int same_flag;
    const bool one_l2 = lq1_same_xcd(sync + 32, blockIdx.x >> 3, &same_flag);
    if (wave == 5) {
        // the loading wave (see the header): row t is in slot t & 3 of the LDS ring in front of the barrier of the step that reads it
        const int l32 = lane & 31;
        auto row = [&](int t) { return *reinterpret_cast<const float4 *>(gates + ((long long)min(t, U - 1) * H + ub + l32) * 4); };
        const float4 a0 = row(0);
        float4 r[4] = {row(1), row(2), row(3), row(4)};      // four steps ahead (first-touch HBM reads: ~2 us, longer than a step)
        if (lane < 32) *reinterpret_cast<float4 *>(&gxr[0][l32 * 4]) = a0;
        __syncthreads();
#pragma unroll 1
        for (int t = 1; t < U; t += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tt = t + k;
                if (tt < U) {
                    if (lane < 32) *reinterpret_cast<float4 *>(&gxr[tt & 3][l32 * 4]) = r[k];
                    r[k] = row(tt + 4);
                    __syncthreads();
                    if (fail_flag) return;
                }
            }
        }
        return;
    }
    __syncthreads();
    if (wave == 4) {      // the polling wave: nothing but the exchange rows in its memory queue
#pragma unroll 1
        for (int t = 1; t < U; ++t) {
            u32x4 row;
            if (lq1_poll_row(rs, ((t - 1) * H + 8 * lane) * 2, row)) *reinterpret_cast<u32x4 *>(hbuf[t & 1] + 8 * lane) = row;
            else if (lane == 0) fail_flag = 1;
            __syncthreads();      // (the buffers alternate: step t + 1's write cannot overtake a wave still reading step t's row)
            if (fail_flag) break;
        }
        if (fail_flag && lane == 0) __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int u0 = ub + 8 * wave;      // this wave's eight units
    // A operand: row m of row block rb = gate (m & 3) of unit u0 + 4 rb + (m >> 2); every B column carries the one batch row, so every lane of
    // a 16-lane group ends up with the four gates of unit u0 + 4 rb + kg
    bf16x8 af[2][KS2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const bf16_t *wrow = whh + ((long long)(m16 & 3) * H + u0 + 4 * rb + (m16 >> 2)) * H;
#pragma unroll
        for (int s = 0; s < KS2; ++s) af[rb][s] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * s + 8 * kg);
    }
    // Every B column is the same row, so the lanes of a 16-lane group all hold the same accumulators: lane (m16 = rb, kg) finishes unit
    // u0 + 4 rb + kg from row block rb - the cell arithmetic runs ONCE per wave-instruction instead of once per row block (a vector
    // instruction is 4 cycles whether 2 or 64 lanes matter; 2 x ~50 of them were a quarter of the step)
    const int rbl = m16 & 1, ulane = 8 * wave + 4 * rbl + kg;      // this lane's unit within the workgroup (lanes m16 >= 2: duplicates, never stored)
    float cprev = 0.f;
    LQ1_INIT();
#pragma unroll 1
    for (int t = 0; t < U; ++t) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        LQ1_STAMP(3);
        if (t > 0) {
            __syncthreads();
            if (fail_flag) break;
        }
        LQ1_STAMP(0);
        const float4 gx = *reinterpret_cast<const float4 *>(&gxr[t & 3][ulane * 4]);
        if (t > 0) {
            const bf16_t *hb = hbuf[t & 1];
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(hb + 32 * s + 8 * kg);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][s], b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][s], b, acc[1], 0, 0, 0);
            }
        }
        const float p0 = rbl ? acc[1][0] : acc[0][0], p1 = rbl ? acc[1][1] : acc[0][1], p2 = rbl ? acc[1][2] : acc[0][2], p3 = rbl ? acc[1][3] : acc[0][3];
        const float gi = sigm(gx.x + p0), gf = sigm(gx.y + p1), gg = tanh_fast(gx.z + p2), go = sigm(gx.w + p3);
        const float cn = gf * cprev + gi * gg;
        cprev = cn;
        const float hv = go * tanh_fast(cn);
        if (m16 < 2) {
            // eight 2-byte stores of one instruction = the wave's 16-byte piece of h_t; a reader that catches it half-written still sees 0xFFFF
            // halves and polls again
            if (one_l2) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)hv), rs, (t * H + ub + ulane) * 2, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)hv), rs, (t * H + ub + ulane) * 2, 0, 16);
            // the step's other outputs (kept for the backward): nobody waits for them
            *reinterpret_cast<float4 *>(gates + ((long long)t * H + ub + ulane) * 4) = make_float4(gi, gf, gg, go);
            c[(long long)t * H + ub + ulane] = cn;
        }
        LQ1_STAMP(2);
    }
    LQ1_DUMP();
}

template <int H>
__global__ __launch_bounds__(384, 1) void lstm_seq1_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ c, const bf16_t *__restrict__ dout,
                                                               bf16_t *dgates /*[U][4H], 0xFFFF-filled*/, const bf16_t *__restrict__ whhT /*[H,4H]*/, unsigned *sync, int U) {
    static_assert(H == 512 && H == LQ1_WGS * 32, "four 16-byte pieces of dG_{t+1} per lane; 16 workgroups x 4 waves x 8 units");
    constexpr int K4 = 4 * H, KH = K4 / 2, KS = KH / 32;
    if (blockIdx.x & 7) return;
    __shared__ __attribute__((aligned(16))) bf16_t gbuf[2][K4];
    // cell operands of the workgroup's 32 units, a four-slot ring filled by the loading wave (see the forward kernel): activated gates, c_t, c_{t-1}, dout
    __shared__ __attribute__((aligned(16))) float opg[4][32 * 4], opc[4][32], opp[4][32];
    __shared__ __attribute__((aligned(16))) bf16_t opd[4][32];
    __shared__ int fail_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kg = lane >> 4;
    const int ub = (blockIdx.x >> 3) * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dgates, 0, U * K4 * 2, 0x00020000);
    if (tid == 0) fail_flag = 0;
    __shared__ int same_flag;
    const bool one_l2 = lq1_same_xcd(sync + 32, blockIdx.x >> 3, &same_flag);
    if (wave == 5) {      // the loading wave: step t's operands are in slot t & 3 in front of the barrier of step t, fetched two steps ahead
        const int l32 = lane & 31;
        auto fetch = [&](int t, float4 &g, float &cn, float &cp, bf16_t &dd) {
            t = max(t, 0);
            g = *reinterpret_cast<const float4 *>(gates + ((long long)t * H + ub + l32) * 4);
            cn = c[(long long)t * H + ub + l32];
            cp = c[(long long)max(t - 1, 0) * H + ub + l32];
            dd = dout[(long long)t * H + ub + l32];
        };
        auto put = [&](int t, const float4 &g, float cn, float cp, bf16_t dd) {
            if (lane < 32) {
                *reinterpret_cast<float4 *>(&opg[t & 3][l32 * 4]) = g;
                opc[t & 3][l32] = cn;
                opp[t & 3][l32] = t > 0 ? cp : 0.f;
                opd[t & 3][l32] = dd;
            }
        };
        float4 g0, g[4]; float cn0, cn[4], cp0, cp[4]; bf16_t d0, d[4];
        fetch(U - 1, g0, cn0, cp0, d0);
#pragma unroll
        for (int k = 0; k < 4; ++k) fetch(U - 2 - k, g[k], cn[k], cp[k], d[k]);      // four steps ahead (see the forward kernel)
        put(U - 1, g0, cn0, cp0, d0);
        __syncthreads();
#pragma unroll 1
        for (int t = U - 2; t >= 0; t -= 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tt = t - k;
                if (tt >= 0) {
                    put(tt, g[k], cn[k], cp[k], d[k]);
                    fetch(tt - 4, g[k], cn[k], cp[k], d[k]);
                    __syncthreads();
                    if (fail_flag) return;
                }
            }
        }
        return;
    }
    __syncthreads();
    if (wave == 4) {      // the polling wave: the row dG_{t+1} is 4 KB = four 16-byte pieces per lane, requested together every round
#pragma unroll 1
        for (int t = U - 2; t >= 0; --t) {
            u32x4 row[4];
            unsigned spins = 0;
            bool ok = true;
            for (;;) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) row[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((t + 1) * K4 + (64 * q + lane) * 8) * 2, 0, 16);
                const bool bad = has_unwritten(row[0]) || has_unwritten(row[1]) || has_unwritten(row[2]) || has_unwritten(row[3]);
                if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
                if (++spins > LQ1_SPINS) { ok = false; break; }
            }
            if (ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4 *>(gbuf[t & 1] + (64 * q + lane) * 8) = row[q];
            } else if (lane == 0) fail_flag = 1;
            __syncthreads();
            if (fail_flag) break;
        }
        if (fail_flag && lane == 0) __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int uw = ub + 8 * wave;      // this wave's eight units
    // A operand: row m = unit uw + (m & 7) of W_hh^T over half (m >> 3) of the reduction range [0, 4H) (gate-major, as dgates rows are laid out);
    // B column n carries half (n & 1) of dG_{t+1}: D[u][0] + D[8 + u][1] is the full sum
    bf16x8 af[KS];
    {
        const bf16_t *wrow = whhT + (long long)(uw + (m16 & 7)) * K4 + (m16 >> 3) * KH;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * s + 8 * kg);
    }
    // The even B columns all carry the first half of dG_{t+1}, the odd ones the second: lane (n = 2 r, kg = k') holds in accumulator register r
    // the first-half sum of unit 4 k' + r, lane (n = 2 r + 1, kg = k' + 2) = that lane + 33 the second-half sum in the same register. Every lane
    // picks "its" register, one cross-lane add finishes the sum, and the cell arithmetic runs ONCE per wave-instruction, a unit per lane
    // (lanes 0, 2, 4, 6, 16, 18, 20, 22), instead of four times in two lanes.
    const int rsel = (m16 >> 1) & 3, ul = 8 * wave + 4 * (kg & 1) + rsel;
    const bool owner = (m16 & 1) == 0 && kg < 2;
    float dcar = 0.f;
    LQ1_INIT();
#pragma unroll 1
    for (int t = U - 1; t >= 0; --t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        const bool last = (t == U - 1);
        LQ1_STAMP(3);
        if (!last) {
            __syncthreads();
            if (fail_flag) break;
        }
        LQ1_STAMP(0);
        const float4 g4 = *reinterpret_cast<const float4 *>(&opg[t & 3][ul * 4]);
        const float cnv = opc[t & 3][ul], cpv = opp[t & 3][ul], dov = (float)opd[t & 3][ul];
        if (!last) {
            const bf16_t *gb = gbuf[t & 1] + (m16 & 1) * KH + 8 * kg;
#pragma unroll
            for (int s = 0; s < KS; s += 2) {      // two chains: a dependent 16x16x32 MFMA issues every 16 cycles
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], *reinterpret_cast<const bf16x8 *>(gb + 32 * s), acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s + 1], *reinterpret_cast<const bf16x8 *>(gb + 32 * s + 32), acc2, 0, 0, 0);
            }
            acc += acc2;
        }
        const float part = rsel == 0 ? acc[0] : rsel == 1 ? acc[1] : rsel == 2 ? acc[2] : acc[3];
        float other;      // (inline asm: hipcc (ROCm 7.2) folded several __builtin_amdgcn_ds_bpermute of different registers into one)
        asm volatile("s_nop 1\n\tds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(other) : "v"(((lane + 33) & 63) * 4), "v"(part));
        const float dhr = part + other;
        const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
        const float tc = tanh_fast(cnv);
        const float dh = dov + (last ? 0.f : dhr);
        const float dc = dh * go * (1.f - tc * tc) + dcar;
        const float dgi = dc * gg * gi * (1.f - gi), dgf = dc * cpv * gf * (1.f - gf), dgg = dc * gi * (1.f - gg * gg), dgo = dh * tc * go * (1.f - go);
        dcar = dc * gf;
        if (owner) {      // 2-byte stores, eight lanes per gate = the wave's 16-byte piece of that gate's row (see the forward kernel)
            const int o = (t * K4 + ub + ul) * 2;
            const unsigned short v0 = __builtin_bit_cast(unsigned short, (bf16_t)dgi), v1 = __builtin_bit_cast(unsigned short, (bf16_t)dgf),
                                 v2 = __builtin_bit_cast(unsigned short, (bf16_t)dgg), v3 = __builtin_bit_cast(unsigned short, (bf16_t)dgo);
            if (one_l2) {
                __builtin_amdgcn_raw_buffer_store_b16(v0, rs, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(v1, rs, o + 2 * H, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(v2, rs, o + 4 * H, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(v3, rs, o + 6 * H, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b16(v0, rs, o, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b16(v1, rs, o + 2 * H, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b16(v2, rs, o + 4 * H, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b16(v3, rs, o + 6 * H, 0, 16);
            }
        }
        LQ1_STAMP(2);
    }
    LQ1_DUMP();
}

static bool lq_seq1(int B, int U, int H) {      // TSASR_LSTM_SEQ1=0: the exchange-group kernels for one utterance too (A/B)
    const char *e = getenv("TSASR_LSTM_SEQ1");      // (read per call: the tests run both routes in one process)
    return (!e || e[0] != '0') && B == 1 && H == 512 && (long long)U * 4 * H * 2 < (1ll << 31);
}
static void fill_async(void *p, size_t bytes, unsigned v, hipStream_t st) {   // bytes % 4 == 0
    const size_t n = bytes / 4;
    fill_words_kernel<<<(unsigned)std::min<size_t>(1024, (n + 255) / 256), 256, 0, st>>>((unsigned *)p, n, v);
}

#define LQ_BR 16   // batch rows per exchange group
// B <= 8 (long-form, one utterance per GPU): groups of 8 rows - the exchange tile a workgroup pulls per step halves (backward: 64 -> 32 KB)
static int lq_br(int B) {
    static const int forced = 0;
    if (forced == 8 || forced == 16) return forced;
    return B <= 8 ? 8 : LQ_BR;
}

static int lq_one_xcd() {      // TSASR_LSTM_XCD=1: every exchange group on one XCD (A/B)
    static const int v = [] { const char *e = getenv("TSASR_LSTM_XCD"); return e && e[0] == '1' ? 1 : 0; }();
    return v;
}

// Zero-fill by a kernel, NOT hipMemsetAsync: a memset node captured into a hipGraph wrote stale host bytes instead of zeros from
// the second replay on (ROCm 7.2, 256-byte fill: the arrival counters then started at garbage and the waits passed early).
__global__ void zero_words_kernel(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static void zero_async(void *p, size_t bytes, hipStream_t st) {   // bytes % 4 == 0
    const size_t n = bytes / 4;
    zero_words_kernel<<<(unsigned)std::min<size_t>(1024, (n + 255) / 256), 256, 0, st>>>((unsigned *)p, n);
}

template <int H, int BR>
static void launch_seq_fwd_br(float *gates, float *c, void *h, const void *whh, int B, int U, char *ws, hipStream_t st) {
    auto kern = lstm_seq_fwd_kernel<H, BR>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
    zero_async(ws, 256, st);
    const int xcd = lq_one_xcd();
    kern<<<dim3((H / LQ_UN) * (xcd ? 8 : 1), cdiv(B, BR)), 256, 96 * 1024, st>>>(gates, c, (bf16_t *)h, (const bf16_t *)whh, (bf16_t *)(ws + 256), (unsigned *)ws, B, U, xcd);
}
template <int H>
static void launch_seq_fwd(float *gates, float *c, void *h, const void *whh, int B, int U, char *ws, hipStream_t st) {
    if (lq_br(B) == 8) launch_seq_fwd_br<H, 8>(gates, c, h, whh, B, U, ws, st);
    else launch_seq_fwd_br<H, LQ_BR>(gates, c, h, whh, B, U, ws, st);
}
template <int H, int BR>
static void launch_seq_bwd_br(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, int B, int U, char *ws, hipStream_t st) {
    auto kern = lstm_seq_bwd_kernel<H, BR>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
    zero_async(ws, 256, st);
    const int xcd = lq_one_xcd();
    kern<<<dim3((H / LQ_UN) * (xcd ? 8 : 1), cdiv(B, BR)), 256, 96 * 1024, st>>>(gates, c, (const bf16_t *)dout, (bf16_t *)dgates, (const bf16_t *)whhT, (bf16_t *)(ws + 256), (unsigned *)ws, B, U, xcd);
}
template <int H>
static void launch_seq_bwd(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, int B, int U, char *ws, hipStream_t st) {
    if (lq_br(B) == 8) launch_seq_bwd_br<H, 8>(gates, c, dout, dgates, whhT, B, U, ws, st);
    else launch_seq_bwd_br<H, LQ_BR>(gates, c, dout, dgates, whhT, B, U, ws, st);
}


// ---- input projection of a ONE-HOT embedding ---------------------------------------------------------------------------------------
// The predictor's Embedding is one-hot (SB/nnet/embedding.py:76-95 consider_as_one_hot: token k -> e_{k-1}, blank -> 0) and frozen, so
// x . W_ih^T is a column of W_ih: gates[b,u,h,g] = b_ih[gH+h] + b_hh[gH+h] + W_ih[gH+h][col(token)], in fp32, gate-minor as the recurrence
// kernels read it. Replaces embedding gather + casts + padded copies + GEMM (17 launches). xp [B*U][Ip] bf16 is the padded one-hot input
// the backward's weight-gradient GEMM contracts with (ones in columns I, I+1: the bias gradient rides in column I).
__global__ __launch_bounds__(256) void lstm_onehot_gates_kernel(const long long *__restrict__ tokens, const float *__restrict__ w_ih,
                                                                 const float *__restrict__ b_ih, const float *__restrict__ b_hh,
                                                                 float *__restrict__ gates, bf16_t *__restrict__ xp, long long BU, int H, int I,
                                                                 int Ip, int blank) {
    const long long n = BU * H;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long row = e / H;
        const int h = (int)(e - row * H);
        const long long tok = tokens[row];
        const int col = (tok == blank || tok < 0) ? -1 : (int)(tok > blank ? tok - 1 : tok);
        float v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = g * H + h;
            v[g] = b_ih[k] + b_hh[k] + ((col >= 0 && col < I) ? w_ih[(long long)k * I + col] : 0.f);
        }
        *reinterpret_cast<float4 *>(gates + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
        if (xp && h < Ip) xp[row * Ip + h] = (bf16_t)((h == col || h == I || h == I + 1) ? 1.f : 0.f);
    }
}

extern "C" {

int tsasr_lstm_cell_fwd(float *gates, float *c, void *h, int B, int U, int H, int t, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(gates && c && h && B > 0 && U > 0 && H > 0 && t >= 0 && t < U, "tsasr_lstm_cell_fwd: bad arguments");
    const int grid = cdiv(B * H, 256);
    if (io_dtype == TSASR_F32) lstm_cell_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (float *)h, B, U, H, t);
    else if (io_dtype == TSASR_BF16) lstm_cell_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (bf16_t *)h, B, U, H, t);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_lstm_cell_fwd");
    return 0;
}

int tsasr_lstm_cell_bwd(const float *gates, const float *c, const void *dout, const float *dh_rec, float *dc_io, void *dgates, int B,
                        int U, int H, int t, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(gates && c && dout && dh_rec && dc_io && dgates && B > 0 && U > 0 && H > 0 && t >= 0 && t < U, "tsasr_lstm_cell_bwd: bad arguments");
    const int grid = cdiv(B * H, 256), last = (t == U - 1);
    if (io_dtype == TSASR_F32) lstm_cell_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (const float *)dout, dh_rec, dc_io, (float *)dgates, B, U, H, t, last);
    else if (io_dtype == TSASR_BF16) lstm_cell_bwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (const bf16_t *)dout, dh_rec, dc_io, (bf16_t *)dgates, B, U, H, t, last);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_lstm_cell_bwd");
    return 0;
}

/* Fused step t of the forward recurrence: gates[:, t] (holding x W_ih^T + biases) += h[:, t-1] . W_hh^T, then the cell update
 * (activated gates, c[:, t], h[:, t] written). whh: bf16 [4H, H]. H % 16 == 0. Launch for t = 0 .. U-1 in order. */
int tsasr_lstm_step_fwd(float *gates, float *c, void *h, const void *whh, int B, int U, int H, int t, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(gates && c && h && whh && B > 0 && U > 0 && H > 0 && H % 16 == 0 && t >= 0 && t < U, "tsasr_lstm_step_fwd: bad arguments (H=%d)", H);
    dim3 grid(H / LS_UN, cdiv(B, 32));
    if (io_dtype == TSASR_BF16) lstm_step_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (bf16_t *)h, (const bf16_t *)whh, B, U, H, t);
    else if (io_dtype == TSASR_F32) lstm_step_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (float *)h, (const bf16_t *)whh, B, U, H, t);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_lstm_step_fwd");
    return 0;
}

/* Fused step t of the backward recurrence (launch for t = U-1 .. 0): dh = dout[:, t] + dgates[:, t+1] . W_hh, cell backward,
 * dgates[:, t] written (io_dtype), dc_io [B,H] fp32 carried. whhT: bf16 [H, 4H] = W_hh transposed. */
int tsasr_lstm_step_bwd(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, float *dc_io, int B, int U,
                        int H, int t, int io_dtype, void *stream) {
    TSASR_CHECK_ARG(gates && c && dout && dgates && whhT && dc_io && B > 0 && U > 0 && H > 0 && H % 16 == 0 && t >= 0 && t < U, "tsasr_lstm_step_bwd: bad arguments");
    dim3 grid(H / LS_UN, cdiv(B, 32));
    if (io_dtype == TSASR_BF16) lstm_step_bwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (const bf16_t *)dout, (bf16_t *)dgates, (const bf16_t *)whhT, dc_io, B, U, H, t);
    else if (io_dtype == TSASR_F32) lstm_step_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(gates, c, (const float *)dout, (float *)dgates, (const bf16_t *)whhT, dc_io, B, U, H, t);
    else TSASR_CHECK_ARG(false, "bad io_dtype %d", io_dtype);
    TSASR_CHECK_LAUNCH("tsasr_lstm_step_bwd");
    return 0;
}

size_t tsasr_lstm_seq_workspace_bytes(int B, int U, int H) {
    const size_t G = (size_t)cdiv(B, 32);
    return 256 + align_up(G * U * 32 * 4 * (size_t)H * sizeof(bf16_t), 256) + align_up((size_t)B * H * sizeof(float), 256);
}

// compute units of the current device (partitioned / CU-masked devices report fewer than the full chip's 256)
static int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cached[dev] = n > 0 ? n : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

// The persistent kernels spin-wait between workgroups: every workgroup of the grid must be resident at once. One workgroup fits a CU
// (96 KB of dynamic LDS, launch bound 1), so the grid may not exceed the CUs THIS device exposes; otherwise the per-step kernels run.
static bool seq_persistent_ok(int B, int H, int io_dtype) {   // sync words: 2 per batch group in a 256-byte block
    static const int force_off = getenv("TSASR_LSTM_PERSISTENT") ? (atoi(getenv("TSASR_LSTM_PERSISTENT")) == 0) : 0;
    return !force_off && io_dtype == TSASR_BF16 && (H == 256 || H == 512) && cdiv(B, lq_br(B)) <= 16 &&
           cdiv(B, lq_br(B)) * (H / LQ_UN) <= device_cu_count();
}

/* gates [B,U,H,4] fp32 = the x-part of the gate pre-activations (both biases included) for one-hot embedded tokens [B,U] (int64):
 * row gH+h of w_ih [4H, I] fp32 at the token's column (token k -> column k-1 above `blank`, k below it; blank -> no column). xp (may be
 * NULL): bf16 [B*U, Ip], Ip >= I + 2, Ip <= H: the one-hot rows with ones in columns I and I+1 (operand of the weight-gradient GEMM). */
int tsasr_lstm_onehot_gates(const long long *tokens, const float *w_ih, const float *b_ih, const float *b_hh, float *gates, void *xp, int B,
                            int U, int H, int I, int Ip, int blank, void *stream) {
    TSASR_CHECK_ARG(tokens && w_ih && b_ih && b_hh && gates && B > 0 && U > 0 && H > 0 && I > 0, "tsasr_lstm_onehot_gates: bad arguments");
    TSASR_CHECK_ARG(!xp || (Ip >= I + 2 && Ip <= H), "tsasr_lstm_onehot_gates: need I + 2 <= Ip <= H (I=%d Ip=%d H=%d)", I, Ip, H);
    const long long n = (long long)B * U * H;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    lstm_onehot_gates_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(tokens, w_ih, b_ih, b_hh, gates, (bf16_t *)xp, (long long)B * U, H, I, Ip, blank);
    TSASR_CHECK_LAUNCH("tsasr_lstm_onehot_gates");
    return 0;
}

/* 1 when tsasr_lstm_seq_fwd/bwd run as ONE persistent launch for this shape on the current device (the workspace's first 256 bytes
 * are then the {arrival counter, error word} pairs), 0 when they loop over the per-step kernels. */
int tsasr_lstm_seq_persistent(int B, int H, int io_dtype) { return seq_persistent_ok(B, H, io_dtype) ? 1 : 0; }

/* The whole forward recurrence (t = 0 .. U-1) of tsasr_lstm_step_fwd. bf16 with H in {256, 512} and B <= 256: ONE persistent launch
 * (csrc/lstm.hip, "Whole-sequence persistent kernels"); otherwise the per-step kernels in a loop. workspace: tsasr_lstm_seq_workspace_bytes. */
int tsasr_lstm_seq_fwd(float *gates, float *c, void *h, const void *whh, int B, int U, int H, int io_dtype, void *workspace,
                       size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(gates && c && h && whh && B > 0 && U > 0 && H > 0 && H % 16 == 0, "tsasr_lstm_seq_fwd: bad arguments (H=%d)", H);
    hipStream_t st = (hipStream_t)stream;
    if (seq_persistent_ok(B, H, io_dtype)) {
        TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_lstm_seq_workspace_bytes(B, U, H), "tsasr_lstm_seq_fwd: workspace too small");
        if (lq_seq1(B, U, H)) {
            zero_async(workspace, 256, st);
            fill_async(h, (size_t)U * H * sizeof(bf16_t), 0xFFFFFFFFu, st);
            lstm_seq1_fwd_kernel<512><<<8 * LQ1_WGS, 384, 0, st>>>(gates, c, (bf16_t *)h, (const bf16_t *)whh, (unsigned *)workspace, U);
        } else if (H == 512) launch_seq_fwd<512>(gates, c, h, whh, B, U, (char *)workspace, st);
        else launch_seq_fwd<256>(gates, c, h, whh, B, U, (char *)workspace, st);
        TSASR_CHECK_LAUNCH("tsasr_lstm_seq_fwd");
        return 0;
    }
    for (int t = 0; t < U; ++t) {
        const int rc = tsasr_lstm_step_fwd(gates, c, h, whh, B, U, H, t, io_dtype, stream);
        if (rc) return rc;
    }
    return 0;
}

/* The whole backward recurrence (t = U-1 .. 0) of tsasr_lstm_step_bwd; same dispatch and workspace as tsasr_lstm_seq_fwd. */
int tsasr_lstm_seq_bwd(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, int B, int U, int H,
                       int io_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(gates && c && dout && dgates && whhT && B > 0 && U > 0 && H > 0 && H % 16 == 0, "tsasr_lstm_seq_bwd: bad arguments");
    TSASR_CHECK_ARG(workspace && workspace_bytes >= tsasr_lstm_seq_workspace_bytes(B, U, H), "tsasr_lstm_seq_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (seq_persistent_ok(B, H, io_dtype)) {
        if (lq_seq1(B, U, H)) {
            zero_async(workspace, 256, st);
            fill_async(dgates, (size_t)U * 4 * H * sizeof(bf16_t), 0xFFFFFFFFu, st);
            lstm_seq1_bwd_kernel<512><<<8 * LQ1_WGS, 384, 0, st>>>(gates, c, (const bf16_t *)dout, (bf16_t *)dgates, (const bf16_t *)whhT, (unsigned *)workspace, U);
        } else if (H == 512) launch_seq_bwd<512>(gates, c, dout, dgates, whhT, B, U, (char *)workspace, st);
        else launch_seq_bwd<256>(gates, c, dout, dgates, whhT, B, U, (char *)workspace, st);
        TSASR_CHECK_LAUNCH("tsasr_lstm_seq_bwd");
        return 0;
    }
    const size_t G = (size_t)cdiv(B, 32);
    float *dc_io = (float *)((char *)workspace + 256 + align_up(G * U * 32 * 4 * (size_t)H * sizeof(bf16_t), 256));
    zero_async(dc_io, (size_t)B * H * sizeof(float), st);
    for (int t = U - 1; t >= 0; --t) {
        const int rc = tsasr_lstm_step_bwd(gates, c, dout, dgates, whhT, dc_io, B, U, H, t, io_dtype, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
