// STATUS (round 5): an experiment that is NOT part of libtsasr_hip.so. Parity-green against the query-major + key-major pair on seven shapes
// (dq / dk / dv / dpk / du / dv_bias within 2e-3 .. 3.5e-3 relative, the bf16 rounding level; tools/scratch/attn_bwd_cmp.py) and through the 62
// attention tests, but SLOWER: 73.8 us at B = 32, T' = 250 (rocprofv3, profiles/r05_notes.md) against 26.8 + 13.4 us for the pair, and the d(pk)
// pass in transposed-source mode 64 us. Why: B*H = 128 workgroups on 256 CUs, one workgroup per CU walking its query blocks in lockstep (two
// barriers and a vmcnt(0) per block, nothing overlapped), 56 spilled VGPRs at 8 waves. lab/attention_fused_dispatch.patch holds the launch-side
// changes of csrc/attention.hip (dispatch + dpk_body's transposed-source mode) that it needs to be linked in again.
//
// Relative-position attention BACKWARD for short sequences as ONE key-major kernel (bf16, Dh = 64, 2 <= T <= 256): dQ, dK, dV, the partial
// sums of d(pos_bias_u / v) and - for the d(pk) pass - scale*dS, without the two [B,H,T,T] tensors (P_d, dS) that the query-major +
// key-major pair of csrc/attention_short.hip writes and reads back (32 + 32 + 36 MB per layer at B = 32, T' = 250: the pair's floor).
// Reference: the backward of RelPosMHAXL.forward's core, vendor/speechbrain/speechbrain/nnet/attention.py:586-633 (rel_shift :468-483).
//
// workgroup = (b, h); wave w owns keys 32w .. 32w+31 (NW = 8 waves for T > 128, 4 for T <= 128): K and V fragments, dK^T and dV^T accumulators
// live in its registers for the whole kernel. The workgroup walks the blocks of 32 queries in lockstep; per block
//   * Q and dO rows (and the forward's dropout keep-bits) arrive by LDS-DMA in a two-stage ring, the whole band of positional rows is resident;
//   * S = Q K^T and dP = dO V^T (key on the lane), G^T = Pband (Q)^T (query on the lane) through a per-wave fp16 tile, read back skewed
//     (BD[i,j] = G[i, j-i+31]) and added by v_fma_mix_f32; p' = exp2(fma(x, scale log2 e, bias_j)) with -lse in the accumulator's initial
//     value and pos_bias_u . k_j, key padding and log2 keep_scale in the per-lane bias; pos_bias_v . p_r is the initial value of G^T;
//   * P_d and scale*dS are the B operands of dV^T += dO^T P_d and dK^T += Q^T dS straight from the accumulator registers (section 3 of the
//     guide, "an accumulator tile as the next MFMA's operand"); dS crosses LDS transposed for dQ += dS K and skewed back for
//     dQ += dG Pband; the wave's share of the block's dQ goes to LDS as bf16 and 512 threads add the 8 shares (fp32) and store dQ;
//   * scale*dS^T leaves as [B,H,j,i] rows for the d(pk) pass (csrc/attention.hip dpk_body, transposed-source mode).
// d(pos_bias_u)[d] = sum_j (sum_i dS[i,j]) k_j[d] from fp32 column sums; d(pos_bias_v) = sum_i dQ[i,:] - d(pos_bias_u).
#include <type_traits>

#include "attn_common.h"

namespace {

template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void tile_store_pair(unsigned addr, unsigned pair) {      // low half -> addr + OFF_LO, high half -> addr + OFF_HI (2 bytes each)
    asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(addr), "v"(pair), "n"(OFF_LO), "n"(OFF_HI) : "memory");
}
__device__ __forceinline__ bf16x8 tr_pair(const char *lo_addr, const char *hi_addr) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(lo_addr));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(hi_addr));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }      // chunk swizzle of the 128-byte-row tiles

}  // namespace

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void relpos_attn_bwd_fused_kernel(
    const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk, const float *__restrict__ bias_u, const float *__restrict__ bias_v,
    const int32_t *__restrict__ key_lens, const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout, const float *__restrict__ lse,
    bf16_t *__restrict__ dqkv, bf16_t *__restrict__ dst_out /*[B,H,Tp(j),Tp(i)] scale*dS transposed*/, float *__restrict__ slab_uv, int nslab,
    int Tp, int Tn, int H, float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *__restrict__ seed_dev,
    const unsigned short *__restrict__ keepbits /*forward's keep-bits [B*H*T][2][8] or NULL*/) {
    constexpr int Dh = 64, TPAD = 32 * NW, NT = 64 * NW;
    constexpr int BAND_OFF = 0, BAND_B = 2 * TPAD * 128, RING_OFF = BAND_B, STAGE = 9216, Q_T = 0, DO_T = 4096, KB_T = 8192;
    constexpr int TAB_OFF = RING_OFF + 2 * STAGE, L_T = 0, DL_T = TPAD * 4, CP_T = 2 * TPAD * 4, U_T = 4 * TPAD * 4, V_T = U_T + 256, RS_T = V_T + 256;
    constexpr int TAB_B = RS_T + TPAD * 4, SCR_OFF = TAB_OFF + TAB_B, SCR_B = 6656, G_T = 0, ST_T = 4608;
    static_assert(SCR_OFF + NW * SCR_B <= 160 * 1024 && TAB_B % 16 == 0, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const bf16_t *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const bf16_t *do_base = dout + ((long long)b * Tn) * D + (long long)h * Dh;
    const bf16_t *o_base = out + ((long long)b * Tn) * D + (long long)h * Dh;
    const bf16_t *p_base = pk + (long long)h * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int NS = (Tn + 31) / 32;                  // blocks of 32 queries
    const int jk = 32 * wave + r;                   // this lane's key
    const int jkc = min(jk, Tn - 1);
    const bool drop = pdrop > 0.f, has_kb = keepbits != nullptr;
    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = drop ? drop_scale16(thr) : 1.f;
    const float c2 = scale * 1.4426950408889634f;
    char *const tab = smem + TAB_OFF;
    char *const scr = smem + SCR_OFF + wave * SCR_B;
    const unsigned scr_a = lds0 + SCR_OFF + wave * SCR_B;

    // ---- ordinary loads first (in flight beside the DMA pieces): this lane's K and V row pieces (B operands: dims 16s + 8hh + [0,8)),
    // and - thread = (query row, half of the dims) - the dO and O pieces behind delta_i = dO_i . O_i
    bf16x8 kf[4], vf[4];
    {
        const bf16_t *krow = q_base + (long long)jkc * row_stride + Dh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8 *>(krow + 16 * s + 8 * hh);
            vf[s] = *reinterpret_cast<const bf16x8 *>(krow + Dh + 16 * s + 8 * hh);
        }
    }
    float dpart = 0.f, lse_row = 0.f;
    const int drow = tid >> 1, dhalf = tid & 1;             // NT / 2 = TPAD rows
    {
        const int rc = min(drow, Tn - 1);
        float d8[4][8], o8[4][8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ld8(do_base + (long long)rc * D + 32 * dhalf + 8 * s, d8[s]);
            ld8(o_base + (long long)rc * D + 32 * dhalf + 8 * s, o8[s]);
        }
        lse_row = lse[((long long)b * H + h) * Tn + rc];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += d8[s][j] * o8[s][j];
    }
    // ---- LDS-DMA: the band (tile row rho <-> table row rho - (TPAD - Tn), clamped), this wave's K block (into its scratch, for the
    // transposed fragments), the pos_bias rows, and the first two stages of the (Q, dO, keep-bits) ring
    const int prow = lane >> 3, pos = lane & 7;
    const unsigned chunk_l = (pos ^ ((((prow >> 1) & 1) << 2) | (prow >> 2) | ((wave & 1) << 1))) << 4;     // S(tile row) for pieces whose first row is 8 * (wave + 8 n)
    const unsigned chunk_0 = (pos ^ ((((prow >> 1) & 1) << 2) | (prow >> 2))) << 4, chunk_1 = chunk_0 ^ 32;   // first row = 8 * even / 8 * odd
    const int qs_b = (int)row_stride * 2, ps_b = D * 2, do_b = D * 2;
    const i32x4 srd_q = make_srd(q_base, (unsigned)(Tn * qs_b) - (unsigned)(h * 3 * Dh * 2));
    const i32x4 srd_do = make_srd(do_base, (unsigned)(Tn * do_b) - (unsigned)(h * Dh * 2));
    const i32x4 srd_p = make_srd(p_base, (unsigned)((2 * Tn - 1) * ps_b) - (unsigned)(h * Dh * 2));
#pragma unroll
    for (int n = 0; n < 2 * TPAD / 8 / NW; ++n) {
        const int pc = wave + NW * n;           // (NW even: the parity of pc is the wave's)
        dma_piece(srd_p, ps_b, chunk_l, pc * 8 - (TPAD - Tn), 0, 2 * Tn - 2, 0u, __builtin_amdgcn_readfirstlane(lds0 + BAND_OFF + pc * 1024), lane);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
        dma_piece(srd_q, qs_b, (n & 1) ? chunk_1 : chunk_0, 32 * wave + 8 * n, 0, Tn - 1, Dh * 2, __builtin_amdgcn_readfirstlane(scr_a + n * 1024), lane);
    if (wave == 1 % NW) {
        at_dma4(bias_u + h * Dh + lane, __builtin_amdgcn_readfirstlane(lds0 + TAB_OFF + U_T));
        at_dma4(bias_v + h * Dh + lane, __builtin_amdgcn_readfirstlane(lds0 + TAB_OFF + V_T));
    }
    auto issue_stage = [&](int s) {         // rows 32 s .. 32 s + 31 of Q and dO (4 pieces each), the keep-bits of those rows (one piece: 32 rows x 32 bytes)
        const unsigned st = lds0 + RING_OFF + (s & 1) * STAGE;
        for (int pc = wave; pc < 9; pc += NW) {
            if (pc < 4) dma_piece(srd_q, qs_b, (pc & 1) ? chunk_1 : chunk_0, 32 * s + 8 * pc, 0, Tn - 1, 0u, __builtin_amdgcn_readfirstlane(st + Q_T + pc * 1024), lane);
            else if (pc < 8) dma_piece(srd_do, do_b, (pc & 1) ? chunk_1 : chunk_0, 32 * s + 8 * (pc - 4), 0, Tn - 1, 0u, __builtin_amdgcn_readfirstlane(st + DO_T + (pc - 4) * 1024), lane);
            else if (has_kb) {
                const long long row0 = ((long long)(b * H + h) * Tn + min(32 * s, Tn - 1)) * 32;       // bytes; rows beyond Tn are never looked at
                const long long lim = ((long long)(b * H + h + 1) * Tn) * 32 - 16;
                at_dma16(reinterpret_cast<const char *>(keepbits) + min(row0 + lane * 16, lim), __builtin_amdgcn_readfirstlane(st + KB_T));
            }
        }
    };
    issue_stage(0);
    if (NS > 1) issue_stage(1);
    // delta and lse tables: the accumulator of S starts at -lse_i / scale (so that exp2(c2 * x) carries exp(-lse_i)); -inf switches a query
    // row beyond the sequence off (its p, dS and everything summed from them are then exact zeros)
    dpart += __shfl_xor(dpart, 1);
    if (dhalf == 0) {
        reinterpret_cast<float *>(tab + L_T)[drow] = drow < Tn ? -lse_row / scale : -INFINITY;
        reinterpret_cast<float *>(tab + DL_T)[drow] = drow < Tn ? dpart * scale / keep_scale : 0.f;
        if (drop && !has_kb) reinterpret_cast<unsigned *>(tab + RS_T)[drow] = attn_row_state((unsigned long long)(b * H + h) * Tn + drow, drop_key(seed));
    }
    wait_vm_barrier<0>();
    // ---- per-lane / per-wave constants
    const char *band = smem + BAND_OFF;
    {   // c'_rho = pos_bias_v . p_rho for every band tile row (thread = row; 2 * TPAD = NT rows)
        const int rho = tid;
        const float *vt = reinterpret_cast<const float *>(tab + V_T);
        float acc = 0.f;
#pragma unroll
        for (int cblk = 0; cblk < 8; ++cblk) {
            const uint4 w = *reinterpret_cast<const uint4 *>(band + rho * 128 + ((cblk ^ swz(rho)) << 4));
            float v8[8];
            ld8(vt + 8 * cblk, v8);
            acc += __uint_as_float(w.x << 16) * v8[0] + __uint_as_float(w.x & 0xffff0000u) * v8[1] + __uint_as_float(w.y << 16) * v8[2] +
                   __uint_as_float(w.y & 0xffff0000u) * v8[3] + __uint_as_float(w.z << 16) * v8[4] + __uint_as_float(w.z & 0xffff0000u) * v8[5] +
                   __uint_as_float(w.w << 16) * v8[6] + __uint_as_float(w.w & 0xffff0000u) * v8[7];
        }
        reinterpret_cast<float *>(tab + CP_T)[rho] = acc;
    }
    const int grp = lane >> 4, gh = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    bf16x8 kt[2][2];        // K^T fragments (A operand of dQ^T += K^T dS^T): rows = dims 32 db + .., k = keys 16 s + 8 hh + e (natural order)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r_lo = 16 * s + 8 * hh + q4, r_hi = r_lo + 4, ch = 4 * db + 2 * gh + (p4 >> 1), inner = (p4 & 1) * 8;
            kt[db][s] = tr_pair(scr + r_lo * 128 + ((ch ^ swz(r_lo)) << 4) + inner, scr + r_hi * 128 + ((ch ^ swz(r_hi)) << 4) + inner);
        }
    float bias_j;           // c2 * (pos_bias_u . k_j) + log2 keep_scale ; -inf for a key beyond the utterance
    {
        const float *ut = reinterpret_cast<const float *>(tab + U_T);
        float cj = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float u8[8];
            ld8(ut + 16 * s + 8 * hh, u8);
#pragma unroll
            for (int e = 0; e < 8; ++e) cj += (float)kf[s][e] * u8[e];
        }
        cj += other_half(cj);
        bias_j = jk < len ? c2 * cj + __log2f(keep_scale) : -INFINITY;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tables written, K blocks read: the scratch tiles may be written

    f32x16 dk_acc[2], dv_acc[2];
    dk_acc[0] = dk_acc[1] = dv_acc[0] = dv_acc[1] = (f32x16){0};
    float cs = 0.f;                                     // sum_i scale*dS[i, j] of this lane's key (each half-wave: its share of the query rows)
    float dqsum[4] = {0.f, 0.f, 0.f, 0.f};              // (reduce threads) column sums of dQ over the query rows this thread has added up
    float one = 1.f;
    asm volatile("" : "+v"(one));
    // dG^T fragment masks: element (band row rl = 16 sp + 8 hh + e, query il = lane & 31) of a block exists iff 0 <= rl + il - 31 < 32
    unsigned dgmask[4][4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl0 = 16 * sp + 8 * hh + 2 * k + r - 31;
            dgmask[sp][k] = ((jl0 >= 0 && jl0 < 32) ? 0xffffu : 0u) | ((jl0 + 1 >= 0 && jl0 + 1 < 32) ? 0xffff0000u : 0u);
        }
    const unsigned skew_a = scr_a + G_T + r * 72 - 280 * hh;     // skewed element (il = i0 + 4 hh, jl = r): + (31 - i0) * 72 + 2 * i0
    const int hq = (r >> 2) & 1, gq = (r & 3) + 4 * (r >> 3);    // where key r sits in the forward's (query-major) keep-bit words: half hq, bit gq
    const bool key_live = 32 * wave < len;                       // (wave-uniform) this key block holds a key of the utterance
    bf16_t *const dst_row = dst_out + ((((long long)b * H + h) * Tp) + jk) * Tp + 8 * hh;

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
        wait_vm_barrier<0>();       // stage s has landed; every wave is through block s - 1 (its dQ shares are added, the scratch is free)
        if (s + 1 < NS && s >= 1) issue_stage(s + 1);
        const char *st = smem + RING_OFF + (s & 1) * STAGE;
        // (wave-uniform) does this (query block, key block) pair hold any unmasked score?
        const int i_lo = 32 * s, i_hi = min(32 * s + 31, Tn - 1);
        const bool any_vis = key_live && (!causal || 32 * wave <= causal_limit(i_hi, causal));
        const bool all_vis = !causal || 32 * wave + 31 <= causal_limit(i_lo, causal);
        if (any_vis) {
            const int rho0 = 32 * (wave - s) + TPAD - 32;          // band tile rows rho0 .. rho0 + 63 <-> local rows jl - il + 31
            // ---- S (from -lse / scale), dP, G^T (from c')
            f32x16 s_acc, dpd = {0};
            {
                const float *lt = reinterpret_cast<const float *>(tab + L_T) + 32 * s + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x = *reinterpret_cast<const float4 *>(lt + 8 * q);
                    s_acc[4 * q] = x.x; s_acc[4 * q + 1] = x.y; s_acc[4 * q + 2] = x.z; s_acc[4 * q + 3] = x.w;
                }
            }
            bf16x8 qf[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qf[k] = *reinterpret_cast<const bf16x8 *>(st + Q_T + r * 128 + (((2 * k + hh) ^ swz(r)) << 4));
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[k], kf[k], s_acc, 0, 0, 0);
                const bf16x8 df = *reinterpret_cast<const bf16x8 *>(st + DO_T + r * 128 + (((2 * k + hh) ^ swz(r)) << 4));
                dpd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[k], dpd, 0, 0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x16 g_acc;
                const float *ct = reinterpret_cast<const float *>(tab + CP_T) + rho0 + 32 * rb + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x = *reinterpret_cast<const float4 *>(ct + 8 * q);
                    g_acc[4 * q] = x.x; g_acc[4 * q + 1] = x.y; g_acc[4 * q + 2] = x.z; g_acc[4 * q + 3] = x.w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = rho0 + 32 * rb + r;
                    const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(band + row * 128 + (((2 * k + hh) ^ swz(row)) << 4));
                    g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qf[k], g_acc, 0, 0, 0);
                }
                const unsigned a = scr_a + G_T + (32 * rb + 4 * hh) * 72 + 2 * r;      // element g <-> local band row (g&3) + 8(g>>2) + 4hh, column r (query)
#define ATF_GSTORE(G) tile_store_pair<(((G) & 3) + 8 * ((G) >> 2)) * 72, (((G) & 3) + 8 * ((G) >> 2)) * 72 + 72>(a, pk_f16(g_acc[G], g_acc[(G) + 1]))
                ATF_GSTORE(0); ATF_GSTORE(2); ATF_GSTORE(4); ATF_GSTORE(6); ATF_GSTORE(8); ATF_GSTORE(10); ATF_GSTORE(12); ATF_GSTORE(14);
#undef ATF_GSTORE
            }
            // ---- BD via the skewed read: element g (query il = i0 + 4 hh, i0 = (g&3) + 8(g>>2)) of this lane's key: tile[(r - il + 31)][il]
            unsigned bd[16];
#define ATF_OFF(G) ((31 - (((G) & 3) + 8 * ((G) >> 2))) * 72 + 2 * (((G) & 3) + 8 * ((G) >> 2)))
            asm volatile(
                "ds_read_u16 %0, %8 offset:%9\n\tds_read_u16 %1, %8 offset:%10\n\tds_read_u16 %2, %8 offset:%11\n\tds_read_u16 %3, %8 offset:%12\n\t"
                "ds_read_u16 %4, %8 offset:%13\n\tds_read_u16 %5, %8 offset:%14\n\tds_read_u16 %6, %8 offset:%15\n\tds_read_u16 %7, %8 offset:%16"
                : "=&v"(bd[0]), "=&v"(bd[1]), "=&v"(bd[2]), "=&v"(bd[3]), "=&v"(bd[4]), "=&v"(bd[5]), "=&v"(bd[6]), "=&v"(bd[7])
                : "v"(skew_a), "n"(ATF_OFF(0)), "n"(ATF_OFF(1)), "n"(ATF_OFF(2)), "n"(ATF_OFF(3)), "n"(ATF_OFF(4)), "n"(ATF_OFF(5)), "n"(ATF_OFF(6)), "n"(ATF_OFF(7))
                : "memory");
            asm volatile(
                "ds_read_u16 %0, %8 offset:%9\n\tds_read_u16 %1, %8 offset:%10\n\tds_read_u16 %2, %8 offset:%11\n\tds_read_u16 %3, %8 offset:%12\n\t"
                "ds_read_u16 %4, %8 offset:%13\n\tds_read_u16 %5, %8 offset:%14\n\tds_read_u16 %6, %8 offset:%15\n\tds_read_u16 %7, %8 offset:%16\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(bd[8]), "=&v"(bd[9]), "=&v"(bd[10]), "=&v"(bd[11]), "=&v"(bd[12]), "=&v"(bd[13]), "=&v"(bd[14]), "=&v"(bd[15])
                : "v"(skew_a), "n"(ATF_OFF(8)), "n"(ATF_OFF(9)), "n"(ATF_OFF(10)), "n"(ATF_OFF(11)), "n"(ATF_OFF(12)), "n"(ATF_OFF(13)), "n"(ATF_OFF(14)), "n"(ATF_OFF(15))
                : "memory");
            // (the first batch's registers are not touched until the second statement's wait has passed: both are consumed below)
            asm volatile("" : "+v"(bd[0]), "+v"(bd[1]), "+v"(bd[2]), "+v"(bd[3]), "+v"(bd[4]), "+v"(bd[5]), "+v"(bd[6]), "+v"(bd[7]));
            float pp[16], ds[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) pp[g] = fast_exp2(__builtin_fmaf(add_h_lo(s_acc[g], bd[g], one), c2, bias_j));      // p * keep_scale
            if (!all_vis) {         // look-ahead mask inside this block
#pragma unroll
                for (int g = 0; g < 16; ++g) pp[g] = (jk > causal_limit(32 * s + (g & 3) + 8 * (g >> 2) + 4 * hh, causal)) ? 0.f : pp[g];
            }
            // ---- dropout keep masks of the 16 (query, this key) pairs; scale*dS = p' (dP & mask) scale - p' delta scale / keep_scale
            const float *dlt = reinterpret_cast<const float *>(tab + DL_T) + 32 * s + 4 * hh;
            float dl[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = *reinterpret_cast<const float4 *>(dlt + 8 * q);
                dl[4 * q] = x.x; dl[4 * q + 1] = x.y; dl[4 * q + 2] = x.z; dl[4 * q + 3] = x.w;
            }
            if (drop) {
                unsigned km[16];
                if (has_kb) {       // word (query row, half hq, key block) of the forward's keep-bits, bit gq
                    const unsigned short *kbt = reinterpret_cast<const unsigned short *>(st + KB_T) + hq * 8 + wave;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const unsigned w = kbt[((g & 3) + 8 * (g >> 2) + 4 * hh) * 16];
                        km[g] = (unsigned)__builtin_amdgcn_sbfe((int)w, gq, 1);
                    }
                } else {
                    const unsigned *rst = reinterpret_cast<const unsigned *>(tab + RS_T) + 32 * s + 4 * hh;
                    const unsigned wi = (unsigned)((wave * 2 + hq) * 8 + (gq >> 1));
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const unsigned y = attn_drop_word(rst[(g & 3) + 8 * (g >> 2)], wi);
                        const int half = (gq & 1) ? (int)y >> 16 : (int)(short)(y & 0xffffu);
                        km[g] = half >= (int)thr - 32768 ? 0xffffffffu : 0u;
                    }
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    ds[g] = pp[g] * __builtin_fmaf(__uint_as_float(__float_as_uint(dpd[g]) & km[g]), scale, -dl[g]);
                    pp[g] = __uint_as_float(__float_as_uint(pp[g]) & km[g]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) ds[g] = pp[g] * __builtin_fmaf(dpd[g], scale, -dl[g]);
            }
            cs += ((ds[0] + ds[1]) + (ds[2] + ds[3])) + ((ds[4] + ds[5]) + (ds[6] + ds[7])) + (((ds[8] + ds[9]) + (ds[10] + ds[11])) + ((ds[12] + ds[13]) + (ds[14] + ds[15])));
            unsigned pdw[8], dsw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pdw[k] = pk_bf16(pp[2 * k], pp[2 * k + 1]);
                dsw[k] = pk_bf16(ds[2 * k], ds[2 * k + 1]);
            }
            // ---- scale*dS^T rows for the d(pk) pass (16-byte stores after one half-wave swap), dS^T and dG^T tiles
            {
                unsigned so[8];
#pragma unroll
                for (int q = 0; q < 4; q += 2)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto rs = __builtin_amdgcn_permlane32_swap(dsw[2 * q + e], dsw[2 * q + 2 + e], false, false);
                        so[2 * q + e] = rs[0]; so[2 * q + 2 + e] = rs[1];
                    }
                if (jk < Tn) {
#pragma unroll
                    for (int q = 0; q < 4; q += 2)      // lanes 0-31: queries 8q .. 8q+7 of the block; lanes 32-63: 8(q+1) .. 8(q+1)+7
                        *reinterpret_cast<uint4 *>(dst_row + 32 * s + 8 * q) = make_uint4(so[2 * q], so[2 * q + 1], so[2 * q + 2], so[2 * q + 3]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)         // dS^T tile [key r][query 8q + 4hh + (0..3)]
                *reinterpret_cast<uint2 *>(scr + ST_T + r * 64 + (8 * q + 4 * hh) * 2) = make_uint2(dsw[2 * q], dsw[2 * q + 1]);
#define ATF_DSTORE(K) tile_store_pair<ATF_OFF(2 * (K)), ATF_OFF(2 * (K) + 1)>(skew_a, dsw[K])
            ATF_DSTORE(0); ATF_DSTORE(1); ATF_DSTORE(2); ATF_DSTORE(3); ATF_DSTORE(4); ATF_DSTORE(5); ATF_DSTORE(6); ATF_DSTORE(7);
#undef ATF_DSTORE
#undef ATF_OFF
            // ---- dV^T += dO^T P_d ; dK^T += Q^T dS (A through the transposing read in the accumulator's row order, B = the pair registers)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const int r_lo = 16 * sp + 4 * hh + q4, r_hi = r_lo + 8, ch = 4 * db + 2 * gh + (p4 >> 1), inner = (p4 & 1) * 8;
                    const int o_lo = r_lo * 128 + ((ch ^ swz(r_lo)) << 4) + inner, o_hi = r_hi * 128 + ((ch ^ swz(r_hi)) << 4) + inner;
                    const bf16x8 a_do = tr_pair(st + DO_T + o_lo, st + DO_T + o_hi), a_q = tr_pair(st + Q_T + o_lo, st + Q_T + o_hi);
                    dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, bf16x8_of(pdw[4 * sp], pdw[4 * sp + 1], pdw[4 * sp + 2], pdw[4 * sp + 3]), dv_acc[db], 0, 0, 0);
                    dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_q, bf16x8_of(dsw[4 * sp], dsw[4 * sp + 1], dsw[4 * sp + 2], dsw[4 * sp + 3]), dk_acc[db], 0, 0, 0);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile stores above (asm / plain) are in LDS before the transposing reads below
            // ---- dQ^T += K^T dS^T (k = key) + Pband^T dG^T (k = band row), both B operands through the transposing read
            f32x16 dq_acc[2];
            dq_acc[0] = dq_acc[1] = (f32x16){0};
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int r_lo = 16 * sp + 8 * hh + q4;
                const bf16x8 bs = tr_pair(scr + ST_T + r_lo * 64 + 32 * gh + 8 * p4, scr + ST_T + (r_lo + 4) * 64 + 32 * gh + 8 * p4);
#pragma unroll
                for (int db = 0; db < 2; ++db) dq_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[db][sp], bs, dq_acc[db], 0, 0, 0);
            }
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int rl = 16 * sp + 8 * hh + q4;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 gl = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(scr + G_T + rl * 72 + 32 * gh + 8 * p4)));
                const u32x2 gh2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(scr + G_T + (rl + 4) * 72 + 32 * gh + 8 * p4)));
                const bf16x8 dgb = bf16x8_of(gl[0] & dgmask[sp][0], gl[1] & dgmask[sp][1], gh2[0] & dgmask[sp][2], gh2[1] & dgmask[sp][3]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int row_lo = rho0 + rl, row_hi = row_lo + 4, ch = 4 * db + 2 * gh + (p4 >> 1), inner = (p4 & 1) * 8;
                    const bf16x8 pa = tr_pair(band + row_lo * 128 + ((ch ^ swz(row_lo)) << 4) + inner, band + row_hi * 128 + ((ch ^ swz(row_hi)) << 4) + inner);
                    dq_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, dgb, dq_acc[db], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tiles' reads have returned: the wave's dQ share may overwrite them
            // ---- this wave's share of the block's dQ (rows = dims, column = query r) -> LDS [query][64 dims] bf16 over the G tile
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2 *>(scr + G_T + r * 128 + (32 * db + 8 * q + 4 * hh) * 2) =
                        make_uint2(pk_bf16(dq_acc[db][4 * q], dq_acc[db][4 * q + 1]), pk_bf16(dq_acc[db][4 * q + 2], dq_acc[db][4 * q + 3]));
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<uint2 *>(scr + G_T + r * 128 + (8 * q + 4 * hh) * 2) = make_uint2(0u, 0u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- dQ of the block = sum of the NW shares (fp32): thread = (query row, 4 dims); NW = 4: two chunks per thread
#pragma unroll
        for (int rep = 0; rep < 512 / NT; ++rep) {
            const int e = tid + rep * NT, il = e >> 4, c = e & 15;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint2 x = *reinterpret_cast<const uint2 *>(smem + SCR_OFF + w * SCR_B + G_T + il * 128 + c * 8);
                a0 += __uint_as_float(x.x << 16); a1 += __uint_as_float(x.x & 0xffff0000u); a2 += __uint_as_float(x.y << 16); a3 += __uint_as_float(x.y & 0xffff0000u);
            }
            const int i = 32 * s + il;
            if (i < Tn) {
                st4(dqkv + ((long long)b * Tn + i) * row_stride + (long long)h * 3 * Dh + 4 * c, a0, a1, a2, a3);
                dqsum[0] += a0; dqsum[1] += a1; dqsum[2] += a2; dqsum[3] += a3;      // (NW = 4: a thread's two rows il, il + 16 share the dims 4c .. 4c+3)
            }
        }
    }
    // ================= epilogue
    __syncthreads();        // every share read; the band and the ring are dead from here
    // dK^T += pos_bias_u[d] * cs_j ; stores (key beyond the utterance: zeros)
    cs += other_half(cs);
    if (jk < Tn) {
        const bool k_live = jk < len;
        const float *ut = reinterpret_cast<const float *>(tab + U_T);
        bf16_t *dkp = dqkv + ((long long)b * Tn + jk) * row_stride + (long long)h * 3 * Dh + Dh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * db + 8 * q + 4 * hh;
                const float4 u4 = *reinterpret_cast<const float4 *>(ut + d);
                float kk[4] = {dk_acc[db][4 * q] + u4.x * cs, dk_acc[db][4 * q + 1] + u4.y * cs, dk_acc[db][4 * q + 2] + u4.z * cs, dk_acc[db][4 * q + 3] + u4.w * cs};
                float vv[4] = {dv_acc[db][4 * q], dv_acc[db][4 * q + 1], dv_acc[db][4 * q + 2], dv_acc[db][4 * q + 3]};
                if (!k_live) { kk[0] = kk[1] = kk[2] = kk[3] = 0.f; vv[0] = vv[1] = vv[2] = vv[3] = 0.f; }
                st4(dkp + d, kk[0], kk[1], kk[2], kk[3]);
                st4(dkp + Dh + d, vv[0], vv[1], vv[2], vv[3]);
            }
    }
    // d(pos_bias_u)[d] = sum_j cs_j k_j[d]: every lane's 32 products (its half of the dims) -> [key][64 dims] fp32 in the dead band, lane d adds a column
    float *red = reinterpret_cast<float *>(smem + BAND_OFF) + wave * 32 * 64;
    static_assert(NW * 32 * 64 * 4 <= BAND_B, "the column-sum tiles fit the dead band");
    const float csl = jk < len ? cs : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; e += 4)
            *reinterpret_cast<float4 *>(red + r * 64 + 16 * s + 8 * hh + e) =
                make_float4(csl * (float)kf[s][e], csl * (float)kf[s][e + 1], csl * (float)kf[s][e + 2], csl * (float)kf[s][e + 3]);
    __builtin_amdgcn_wave_barrier();
    float du_w = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) du_w += red[((k + lane) & 31) * 64 + lane];
    __syncthreads();
    float *fin = reinterpret_cast<float *>(smem + BAND_OFF);        // [NW][64] du shares, then [32][64] dQ column sums
    fin[wave * 64 + lane] = du_w;
    float *dqs = fin + NW * 64;
    {
        // (reduce threads) thread e = tid (+ NT) owned (query row il = e >> 4, dims 4c .. 4c+3); NW = 4 threads own two rows il, il + 16 of the same dims
        const int il = (tid >> 4) & 31, c = tid & 15;
        *reinterpret_cast<float4 *>(dqs + il * 64 + 4 * c) = make_float4(dqsum[0], dqsum[1], dqsum[2], dqsum[3]);
    }
    __syncthreads();
    if (tid < 64) {
        float du = 0.f, dq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) du += fin[w * 64 + tid];
        constexpr int NR = NT / 16;          // rows of dQ column sums that were written (32 for NW = 8, 16 for NW = 4)
#pragma unroll
        for (int il = 0; il < NR; ++il) dq += dqs[il * 64 + tid];
        float *slab = slab_uv + (((long long)b * nslab) * H + h) * 128;        // part 0 of this utterance; parts 1 .. nslab-1 are zeroed below
        slab[tid] = du;
        slab[64 + tid] = dq - du;
    }
    for (int e = tid; e < (nslab - 1) * 128; e += NT) slab_uv[(((long long)b * nslab + 1 + e / 128) * H + h) * 128 + (e & 127)] = 0.f;
}

extern "C" {

/* Fused short-sequence backward (bf16, Dh = 64, 2 <= T <= 256): called by tsasr_relpos_attn_bwd (csrc/attention.hip) in place of the query-major
 * and key-major passes; dst = scale*dS TRANSPOSED ([B,H,Tp,Tp], key-major rows) for the d(pk) pass; slab: nslab rows per utterance, row 0 filled. */
int tsasr_attn_fused_bwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *out,
                         const void *dout, const float *lse, void *dqkv, void *dst, float *slab, int nslab, int Tp, int B, int T, int H, float scale,
                         int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, const void *keepbits, hipStream_t st) {
    if (T > 128) {
        constexpr int NW = 8, LDSS = 2 * 32 * NW * 128 + 2 * 9216 + (5 * 32 * NW * 4 + 512) + NW * 6656;
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_fused_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_bwd_fused_kernel<8><<<dim3(B * H), 64 * NW, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, (const bf16_t *)out,
                                                                           (const bf16_t *)dout, lse, (bf16_t *)dqkv, (bf16_t *)dst, slab, nslab, Tp, T, H, scale, causal,
                                                                           pdrop, seed, seed_dev, (const unsigned short *)keepbits);
    } else {
        constexpr int NW = 4, LDSS = 2 * 32 * NW * 128 + 2 * 9216 + (5 * 32 * NW * 4 + 512) + NW * 6656;
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_bwd_fused_kernel<4><<<dim3(B * H), 64 * NW, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, (const bf16_t *)out,
                                                                           (const bf16_t *)dout, lse, (bf16_t *)dqkv, (bf16_t *)dst, slab, nslab, Tp, T, H, scale, causal,
                                                                           pdrop, seed, seed_dev, (const unsigned short *)keepbits);
    }
    return 0;
}

}  // extern "C"
