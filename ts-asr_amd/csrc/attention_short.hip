// Relative-position attention for SHORT sequences (2 <= T <= 256, bf16, Dh = 64): the mixture encoder's T' = 250 and the speaker
// encoder's T' = 125 at BASELINE configs[1]. Same arithmetic, masks, dropout stream and outputs as the streaming kernels of
// csrc/attention.hip (reference: RelPosMHAXL.forward, vendor/speechbrain/speechbrain/nnet/attention.py:586-633, rel_shift :468-483).
//
// What is different from the streaming kernels (and from round 3's relpos_attn_fwd_short_kernel, which this file replaces):
//   * everything a workgroup needs comes in by LDS-DMA in GROUPS that follow the order of use (query rows + first key block + the band
//     rows the first score blocks touch; then, per 32-key step, the next K block, the V block of the step before and 32 more band rows):
//     the first MFMA starts when ~half of the bytes have landed, the rest lands behind the score blocks (one counted s_waitcnt vmcnt +
//     s_barrier per step; groups are issued two steps ahead). Round 3 waited for all 112 KB first: 31 % of the kernel.
//   * the vector work per score element is cut to the bone (it, not the MFMAs, set the pace: 24 + 19 % of the cycles):
//       - G (band product) tiles cross LDS as fp16 PAIRS: v_cvt_pk_f16_f32 on the way in, ds_read_u16_d16 / _d16_hi on the way out, and
//         the skewed value is added to the AC accumulator by v_fma_mix_f32 (no conversion instruction);
//       - exp(scale * x - m) = exp2(fma(x, scale * log2 e, -m')) - one fma, one v_exp_f32;
//       - the running maximum is only raised when a row's new maximum exceeds it by 2^6 (wave-uniform branch): the rescale of the
//         32 O accumulators and of l happens once or twice per row instead of once per 32 keys (p <= 2^6: no loss in fp32 / bf16);
//       - row sums are kept per half-wave and combined once at the end;
//       - dropout: one 24-bit-multiply word per two keys (csrc/attn_common.h), pair masks by two packed 16-bit operations, applied
//         to the bf16 PAIR after the conversion (one AND per two elements); 1 / (1 - p) is folded into the final 1 / l;
//       - P.V of a step runs at the head of the NEXT step, behind that step's AC / G MFMAs (its V block arrives with that group).
//   * the dropout keep-bits the backward reads (tsasr_relpos_attn_keepbits) leave as ONE 8-byte store per lane.
#include <algorithm>
#include <type_traits>

#include "attn_common.h"


// =====================================================================================================================
// Forward. workgroup = (b, h, QH queries), 8 waves = (query blocks of 32) x (key parts): 4 x 2 for QH = 128 (T > 128: keys padded to
// 256, two parts of 128), 2 x 4 for QH = 64 (keys padded to 128, four parts of 32); a lane owns ONE query (accumulators: rows = keys /
// band rows / head dims, column = query), the key parts of a query block are merged through LDS at the end.
// =====================================================================================================================
// CHUNK (QH = 128, T > 256: long sequences in small batches, BASELINE configs[4]'s T' = 4000): the same workgroup against ONE chunk of 256 keys
// (keys j0 = 256 * part ..), one workgroup per (b, h, 128 queries, chunk that the tile's causal / length limit reaches); it leaves its
// un-normalised (O, m, l) in part_o / part_ml exactly as relpos_attn_fwd_kernel's key parts do, and relpos_attn_merge_kernel combines the
// chunks of a query. ~1100 live workgroups of equal size at T' = 4000 instead of 384 of very unequal size.
// the live (query tile, chunk) pairs of one (b, h), chunk-major: chunk p is met by the query tiles first_qt[p] .. nqt - 1 (host-computed from
// the look-ahead rule with every key present; shorter utterances make some of them return at once); passed to the kernel by value
struct AttnChunkPlan {
    int per_pair, nparts, base[65], first_qt[64];
};

template <int QH, bool CHUNK>
__device__ __forceinline__ void relpos_attn_fwd_short2_item(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk,
                                                            const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                            const int32_t *__restrict__ key_lens, bf16_t *__restrict__ out,
                                                            float *__restrict__ lse, int Tn, int H, float scale, int causal,
                                                            float pdrop, unsigned long long seed,
                                                            const unsigned long long *__restrict__ seed_dev,
                                                            unsigned short *__restrict__ keepbits /*[B*H*T][2][8] or NULL*/,
                                                            unsigned bid, unsigned nbid, const AttnChunkPlan *__restrict__ plan, float *__restrict__ part_o,
                                                            float *__restrict__ part_ml) {
    constexpr int Dh = 64, NQB = QH / 32, NKP = 8 / NQB, TPAD = 2 * QH, KP = TPAD / NKP, NSUB = KP / 32, NB = QH + TPAD;
    static_assert(!CHUNK || QH == 128, "chunks are 256 keys against 128 queries");
    constexpr int K_OFF = 0, V_OFF = TPAD * 128, P_OFF = 2 * TPAD * 128, G_OFF = P_OFF + NB * 128;
    constexpr int Q_OFF = G_OFF, UV_OFF = G_OFF + QH * 128;      // the query rows and the two bias rows live in the G scratch until they are read
    static_assert(QH * 128 + 512 <= 8 * 6144, "query staging fits the G scratch");
    // band rows of group 0: what step 0 needs (NB - 32 (NSUB - 1) rows), rounded up to whole rounds of 8 pieces; the rest rides in group 1
    constexpr int NBA = NSUB == 1 ? NB : (((NB - 32 * (NSUB - 1)) / 8 + 7) / 8) * 64, NBB = NB - NBA;
    static_assert(QH / 8 % 8 == 0 && NKP * 4 % 8 == 0 && NBA / 8 % 8 == 0 && NBB / 8 % 8 == 0 && NBA <= NB, "every round of 8 pieces has one kind");
    constexpr int G0_W = (QH / 8 + NKP * 4 + NBA / 8) / 8 + 2, G1_W = (2 * NKP * 4 + NBB / 8) / 8, GS_W = 2 * NKP * 4 / 8, GL_W = NKP * 4 / 8;   // pieces per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    // blockIdx.x -> (query tile, (b, h)): the query tiles of one (b, h) get block ids 8 apart, i.e. the same XCD under round-robin
    // dispatch - they fetch the same K / V rows (speed only: any placement is correct)
    const int nqt = (Tn + QH - 1) / QH, npair = nbid / nqt;
    const int nparts = CHUNK ? plan->nparts : 1;
    int qt, pair, part = 0;
    if (CHUNK) {            // neighbouring items: the same chunk of keys against consecutive query tiles
        pair = bid / plan->per_pair;
        const int rem = bid - pair * plan->per_pair;
        while (plan->base[part + 1] <= rem) ++part;
        qt = plan->first_qt[part] + (rem - plan->base[part]);
    } else if ((npair & 7) == 0) {
        const int k = bid >> 3;
        qt = k % nqt;
        pair = (k / nqt) * 8 + (bid & 7);
    } else {
        qt = bid % nqt;
        pair = bid / nqt;
    }
    const int b = pair / H, h = pair % H, i0 = qt * QH, j0 = CHUNK ? part * TPAD : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int qb = wave % NQB, kp = wave / NQB;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const bf16_t *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const bf16_t *p_base = pk + (long long)h * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    if (CHUNK) {            // a chunk beyond what the tile's last query attends: no such part for the merge (workgroup-uniform, before any barrier)
        const int i_last = min(i0 + QH - 1, Tn - 1 + QH);
        const int j_end = causal ? min(len, causal_limit(i_last, causal) + 1) : len;
        if (j0 >= j_end) return;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int r_lo = (Tn - 1) - (i0 + QH - 1) + j0; // band row R of the workgroup <-> table row r_lo + R (clamped: out-of-table rows only meet masked keys)

#ifdef AT_PROFILE
    long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#define ATS_STAMP(i) do { const long long n_ = clock64(); sacc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define ATS_STAMP(i)
#endif
    // DMA groups (pieces of 8 rows; piece index = wave + 8 n, so every n of a group has ONE kind of piece for all waves):
    //   g0 = Q, K block 0 of every part, band rows [0, NBA), the two bias rows; g1 = V block 0, K block 1, band rows [NBA, NB);
    //   g(s) = V block s-1, K block s; g(NSUB) = V block NSUB-1.
    const int prow = lane >> 3, pos = lane & 7;
    const unsigned par = (wave & 1) * 4;          // (row0 >> 1) & 4 of this wave's pieces: piece rows are 8 * (wave + 8 n + const * 2)
    const int qs_b = (int)row_stride * 2, ps_b = D * 2;
    const unsigned voff_q = (pos ^ (prow >> 1) ^ par) << 4;                 // K / Q / band tiles: swizzle (row >> 1) & 7
    const unsigned voff_v = (pos ^ (((prow >> 1) & 1) << 2)) << 4;          // V tile: swizzle ((row >> 1) & 1) << 2
    const unsigned voff_p = voff_q;
    const i32x4 srd_q = make_srd(q_base, (unsigned)(Tn * qs_b) - (unsigned)(h * 3 * Dh * 2));
    const i32x4 srd_p = make_srd(p_base, (unsigned)((2 * Tn - 1) * ps_b) - (unsigned)(h * Dh * 2));
    auto issue_group = [&](int g) {      // g is a compile-time constant at every call site (unrolled)
        if (g == 0) {
#pragma unroll
            for (int n = 0; n < G0_W - 2; ++n) {
                const int pc = wave + 8 * n;
                if (8 * n < QH / 8)
                    dma_piece(srd_q, qs_b, voff_q, i0 + pc * 8, 0, Tn - 1, 0u, __builtin_amdgcn_readfirstlane(lds0 + Q_OFF + pc * 1024), lane);
                else if (8 * n < QH / 8 + NKP * 4) {
                    const int e = pc - QH / 8, row = (e >> 2) * KP + (e & 3) * 8;
                    dma_piece(srd_q, qs_b, voff_q, j0 + row, 0, Tn - 1, Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + K_OFF + row * 128), lane);
                } else {
                    const int row = (pc - QH / 8 - NKP * 4) * 8;
                    dma_piece(srd_p, ps_b, voff_p, r_lo + row, 0, 2 * Tn - 2, 0u, __builtin_amdgcn_readfirstlane(lds0 + P_OFF + row * 128), lane);
                }
            }
            at_dma4(bias_u + h * Dh + lane, __builtin_amdgcn_readfirstlane(lds0 + UV_OFF));          // (every wave: identical bytes)
            at_dma4(bias_v + h * Dh + lane, __builtin_amdgcn_readfirstlane(lds0 + UV_OFF + 256));
        } else if (g < NSUB) {
#pragma unroll
            for (int n = 0; n < (g == 1 ? G1_W : GS_W); ++n) {
                const int pc = wave + 8 * n;
                if (8 * n < NKP * 4) {
                    const int row = (pc >> 2) * KP + 32 * (g - 1) + (pc & 3) * 8;
                    dma_piece(srd_q, qs_b, voff_v, j0 + row, 0, Tn - 1, 2 * Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + V_OFF + row * 128), lane);
                } else if (8 * n < 2 * NKP * 4) {
                    const int e = pc - NKP * 4, row = (e >> 2) * KP + 32 * g + (e & 3) * 8;
                    dma_piece(srd_q, qs_b, voff_q, j0 + row, 0, Tn - 1, Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + K_OFF + row * 128), lane);
                } else {
                    const int row = NBA + (pc - 2 * NKP * 4) * 8;
                    dma_piece(srd_p, ps_b, voff_p, r_lo + row, 0, 2 * Tn - 2, 0u, __builtin_amdgcn_readfirstlane(lds0 + P_OFF + row * 128), lane);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < GL_W; ++n) {
                const int pc = wave + 8 * n;
                const int row = (pc >> 2) * KP + 32 * (NSUB - 1) + (pc & 3) * 8;
                dma_piece(srd_q, qs_b, voff_v, j0 + row, 0, Tn - 1, 2 * Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + V_OFF + row * 128), lane);
            }
        }
    };
    // every group is issued up front (a wave has nothing else to do until group 0 has landed; the pieces of the later groups then
    // arrive behind the first steps); pieces per wave still in flight when group g must have landed:
    constexpr int W_AFTER0 = (NSUB > 1 ? G1_W + (NSUB - 2) * GS_W : 0) + GL_W;
    issue_group(0);
    issue_group(1);
    if constexpr (NSUB > 2) issue_group(2);
    if constexpr (NSUB > 3) issue_group(3);
    if constexpr (NSUB > 1) issue_group(NSUB);
    ATS_STAMP(0);   // DMA issue
    wait_vm_barrier<W_AFTER0>();                         // group 0 is in LDS
    ATS_STAMP(1);   // wait for group 0

    // ---- this lane's query: Q + u, Q + v as B operands (dims 16s + 8hh + [0,8))
    const int iq = i0 + 32 * qb + r;
    bf16x8 qu[4], qv[4];
    {
        const char *q_lds = smem + Q_OFF + (32 * qb + r) * 128;
        const int sw = ((32 * qb + r) >> 1) & 7;
        const float *u_lds = reinterpret_cast<const float *>(smem + UV_OFF), *v_lds2 = u_lds + 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 qw = *reinterpret_cast<const uint4 *>(q_lds + (((2 * s + hh) ^ sw) << 4));
            float q8[8], u8[8], v8[8];
            q8[0] = __uint_as_float(qw.x << 16); q8[1] = __uint_as_float(qw.x & 0xffff0000u);
            q8[2] = __uint_as_float(qw.y << 16); q8[3] = __uint_as_float(qw.y & 0xffff0000u);
            q8[4] = __uint_as_float(qw.z << 16); q8[5] = __uint_as_float(qw.z & 0xffff0000u);
            q8[6] = __uint_as_float(qw.w << 16); q8[7] = __uint_as_float(qw.w & 0xffff0000u);
            ld8(u_lds + 16 * s + 8 * hh, u8);
            ld8(v_lds2 + 16 * s + 8 * hh, v8);
            qu[s] = bf16x8_of(pk_bf16(q8[0] + u8[0], q8[1] + u8[1]), pk_bf16(q8[2] + u8[2], q8[3] + u8[3]), pk_bf16(q8[4] + u8[4], q8[5] + u8[5]),
                              pk_bf16(q8[6] + u8[6], q8[7] + u8[7]));
            qv[s] = bf16x8_of(pk_bf16(q8[0] + v8[0], q8[1] + v8[1]), pk_bf16(q8[2] + v8[2], q8[3] + v8[3]), pk_bf16(q8[4] + v8[4], q8[5] + v8[5]),
                              pk_bf16(q8[6] + v8[6], q8[7] + v8[7]));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has its query: the G scratch may be written
    ATS_STAMP(1);

    f32x16 o_acc[2];
    o_acc[0] = (f32x16){0};
    o_acc[1] = (f32x16){0};
    const float c2 = scale * 1.4426950408889634f;          // exp(scale * x) = exp2(c2 * x)
    const float thr_raw = AT2_THR_LOG2 / c2;
    float one = 1.f;
    asm volatile("" : "+v"(one));                          // (see add_h_lo)
    float m_run = -1e30f, l_run = 0.f;                     // (m in units of the raw score; l: this HALF-wave's share of the row sum)
    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = drop_scale16(thr);
    const bool drop = pdrop > 0.f;
    const unsigned row_state = attn_row_state((unsigned long long)(b * H + h) * Tn + iq, drop_key(seed));
    const int lim_q = causal ? causal_limit(iq, causal) : 0x3fffffff;
    const int lim_blk = causal ? causal_limit(min(i0 + 32 * qb + 31, Tn - 1), causal) : 0x3fffffff;   // last key any query of this wave attends
    const int j_lim = min(len - 1, lim_q) - j0; // last key this lane's query attends (index inside the chunk) ...
    int j_all = j_lim;                          // ... and the last one EVERY query of the wave attends
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) j_all = min(j_all, __shfl_xor(j_all, o));
    j_all = __builtin_amdgcn_readfirstlane(j_all);
    const char *k_lds = smem + K_OFF, *v_lds = smem + V_OFF, *p_lds = smem + P_OFF;
    const int fr_swz = (r >> 1) & 7;   // fragment rows are 32-aligned + r: the swizzle term of a b128 fragment read is the lane's own
    const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;

    // G tiles of this wave: three slots of [32 band rows][32 queries] fp16. The window of a step (64 band rows: block `sub` below block
    // `sub + 1`) must be CONTIGUOUS for the skewed read (its addresses are base + immediate): even steps read slots [0][1], odd steps
    // slots [1][2], where slot 2 duplicates slot 0 (a block computed at an odd step is stored twice).
    const unsigned g_addr = lds0 + G_OFF + wave * 6144;
    auto g_block = [&](int Rblk, unsigned dst0, bool dup) {   // G^T rows Rblk .. Rblk+31 = Pband . (Q+v)^T -> fp16 at byte offset dst0 (and dst0 + 4096)
        f32x16 g_acc = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(p_lds + (Rblk + r) * 128 + (((2 * s + hh) ^ fr_swz) << 4));
            g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qv[s], g_acc, 0, 0, 0);
        }
        const unsigned a = g_addr + dst0 + (4 * hh) * 64 + 2 * r;      // accumulator element g <-> band row (g&3) + 8(g>>2) + 4hh, column r
#define AT2_GSTORE(G) g_store_pair<(((G) & 3) + 8 * ((G) >> 2)) * 64>(a, pk_f16(g_acc[G], g_acc[(G) + 1]), dup)
        AT2_GSTORE(0); AT2_GSTORE(2); AT2_GSTORE(4); AT2_GSTORE(6); AT2_GSTORE(8); AT2_GSTORE(10); AT2_GSTORE(12); AT2_GSTORE(14);   // rows of g and g + 1 are adjacent
#undef AT2_GSTORE
    };
    const int j_first = kp * KP;
    int j_last = min(min((kp + 1) * KP, len - j0), lim_blk + 1 - j0);    // keys [j_first, j_last) of the chunk are live for this wave
    if (i0 + 32 * qb >= Tn) j_last = j_first;                   // a query block beyond the sequence
    const int nsub = j_last > j_first ? (j_last - j_first + 31) / 32 : 0;      // wave-uniform
    const int Rb0 = j_first - 32 * qb + QH - 32;                // band block of step 0 (rows Rb0 .. Rb0 + 31), the new block of step s: Rb0 + 32 (s + 1)
    if (nsub > 0) g_block(Rb0, 0, false);

    bf16x8 pb[2];                       // probabilities of the previous step (B operand of its P.V, run at the head of the next step)
    unsigned kbits[(NSUB + 1) / 2];     // keep-bits of the steps, two per register
#pragma unroll
    for (int i = 0; i < (NSUB + 1) / 2; ++i) kbits[i] = 0;
    auto pv = [&](int sub) {            // O^T += V^T . P^T ; A = V^T through the transposing read (k order of pb = accumulator row order)
        const int jb = j_first + 32 * sub;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row_lo = jb + 16 * s + 4 * hh + q4, row_hi = row_lo + 8, col = 32 * db + 16 * mhalf + 4 * p4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4 *)(v_lds + row_lo * 128 + (((col >> 3) ^ (((row_lo >> 1) & 1) << 2)) << 4) + (col & 7) * 2));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4 *)(v_lds + row_hi * 128 + (((col >> 3) ^ (((row_hi >> 1) & 1) << 2)) << 4) + (col & 7) * 2));
                bf16x8 va;
                va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3]; va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
                o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[s], o_acc[db], 0, 0, 0);
            }
    };
    auto step = [&](auto sub_tag) {
        constexpr int sub = decltype(sub_tag)::value;
        // group `sub` has landed (K block, band rows, the V block of the step before); group sub + 2 goes out behind the barrier
        if constexpr (sub > 0) {
            // ONE more rendezvous: everything has landed before step 1. A barrier per step kept the two waves of a SIMD in the same phase
            // (both in their MFMAs, then both in their softmax: nothing overlapped, 29 % of the cycles were spent at the barriers);
            // behind this one the waves run free and drift into each other's shadows.
            if constexpr (sub == 1) wait_vm_barrier<0>();
        }
        ATS_STAMP(5);   // group waits + barriers of the steps
        const bool live = sub < nsub;                    // wave-uniform
        const int jb = j_first + 32 * sub;
        f32x16 s_acc = {0};
        if (live) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(k_lds + (jb + r) * 128 + (((2 * s + hh) ^ fr_swz) << 4));
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qu[s], s_acc, 0, 0, 0);
            }
            g_block(Rb0 + 32 * (sub + 1), (sub & 1) ? 0u : 2048u, (sub & 1) != 0);
        }
        if constexpr (sub > 0) {
            if (sub - 1 < nsub) pv(sub - 1);
        }
        ATS_STAMP(2);   // DMA issue of the group two steps ahead, AC + G MFMAs, G store, P.V of the step before
        if (!live) return;
        // ---- raw scores of this lane's query: 16 keys jb + (g&3) + 8(g>>2) + 4hh ; BD via the skewed read of the fp16 window
        // (16 registers, one value each in the low half: two d16 loads in flight into the two halves of ONE register lose the first one)
        unsigned bd[16];
        {
            const unsigned a = g_addr + ((sub & 1) ? 2048u : 0u) + (31 - r + 4 * hh) * 64 + 2 * r;
            asm volatile(
                "ds_read_u16 %0, %16 offset:0\n\tds_read_u16 %1, %16 offset:64\n\t"
                "ds_read_u16 %2, %16 offset:128\n\tds_read_u16 %3, %16 offset:192\n\t"
                "ds_read_u16 %4, %16 offset:512\n\tds_read_u16 %5, %16 offset:576\n\t"
                "ds_read_u16 %6, %16 offset:640\n\tds_read_u16 %7, %16 offset:704\n\t"
                "ds_read_u16 %8, %16 offset:1024\n\tds_read_u16 %9, %16 offset:1088\n\t"
                "ds_read_u16 %10, %16 offset:1152\n\tds_read_u16 %11, %16 offset:1216\n\t"
                "ds_read_u16 %12, %16 offset:1536\n\tds_read_u16 %13, %16 offset:1600\n\t"
                "ds_read_u16 %14, %16 offset:1664\n\tds_read_u16 %15, %16 offset:1728\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(bd[0]), "=&v"(bd[1]), "=&v"(bd[2]), "=&v"(bd[3]), "=&v"(bd[4]), "=&v"(bd[5]), "=&v"(bd[6]), "=&v"(bd[7]),
                  "=&v"(bd[8]), "=&v"(bd[9]), "=&v"(bd[10]), "=&v"(bd[11]), "=&v"(bd[12]), "=&v"(bd[13]), "=&v"(bd[14]), "=&v"(bd[15])
                : "v"(a)
                : "memory");
        }
        float t[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) t[g] = add_h_lo(s_acc[g], bd[g], one);
        if (jb + 31 > j_all) {          // (wave-uniform) some key of the block is masked for some query: key padding, look-ahead mask, T % 32
#pragma unroll
            for (int g = 0; g < 16; ++g) t[g] = (jb + (g & 3) + 8 * (g >> 2) + 4 * hh > j_lim) ? -INFINITY : t[g];
        }
        float mx = fmaxf(fmaxf(t[0], t[1]), t[2]);
#pragma unroll
        for (int g = 3; g < 15; g += 2) mx = fmaxf(fmaxf(mx, t[g]), t[g + 1]);
        mx = fmaxf(mx, t[15]);
        mx = fmaxf(mx, other_half(mx));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + thr_raw) != 0) {      // rare after the first block: raise the running maximum
            const float m_new = fmaxf(m_run, mx), alpha = fast_exp2((m_run - m_new) * c2);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; ++g) o_acc[db][g] *= alpha;
        }
        const float mc = m_run * c2;
        float p[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) p[g] = fast_exp2(__builtin_fmaf(t[g], c2, -mc));
        l_run += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])) + (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
        unsigned pw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pw[k] = pk_bf16(p[2 * k], p[2 * k + 1]);
        if (drop) {
            unsigned km[8];
            attn_pair_masks(row_state, (j0 + jb) >> 5, hh, thr, km);
#pragma unroll
            for (int k = 0; k < 8; ++k) pw[k] &= km[k];
            kbits[sub >> 1] |= attn_keep_bits(km) << (16 * (sub & 1));
        }
        pb[0] = bf16x8_of(pw[0], pw[1], pw[2], pw[3]);
        pb[1] = bf16x8_of(pw[4], pw[5], pw[6], pw[7]);
        ATS_STAMP(3);   // skewed read, softmax, dropout, P fragments
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (NSUB > 1) step(std::integral_constant<int, 1>{});
    if constexpr (NSUB > 2) step(std::integral_constant<int, 2>{});
    if constexpr (NSUB > 3) step(std::integral_constant<int, 3>{});
    static_assert(NSUB <= 4, "steps are unrolled by hand");
    if constexpr (NSUB == 1) wait_vm_barrier<0>();      // the V block (NSUB > 1: landed before step 1)
    ATS_STAMP(5);
    if (NSUB - 1 < nsub) pv(NSUB - 1);
    ATS_STAMP(4);   // last P.V
    // keep-bits for the backward: [row][hh][8 blocks of 32 keys] u16; this wave's NSUB blocks are consecutive
    if (!CHUNK && drop && keepbits && iq < Tn && nsub > 0) {
        unsigned short *kb = keepbits + (((size_t)(b * H + h) * Tn + iq) * 2 + hh) * 8 + kp * NSUB;
        if constexpr (NSUB == 4) *reinterpret_cast<uint2 *>(kb) = make_uint2(kbits[0], kbits[1]);
        else if constexpr (NSUB == 2) *reinterpret_cast<unsigned *>(kb) = kbits[0];
        else *kb = (unsigned short)kbits[0];
    }
    // ---- merge the key parts of each query block: parts kp > 0 hand (m, l, O) to part 0 through LDS (K / V / band are dead now)
    l_run += other_half(l_run);
    __syncthreads();
    ATS_STAMP(6);   // barrier before the merge
    float *mrg = reinterpret_cast<float *>(smem);   // [(kp - 1) * NQB + qb][64 dims x 32 queries | m[32] | l[32]]
    constexpr int MSZ = 64 * 32 + 64;
    static_assert((NKP - 1) * NQB * MSZ * 4 <= G_OFF, "merge buffers fit the dead K / V / band region");
    if (kp > 0) {
        float *mine = mrg + ((kp - 1) * NQB + qb) * MSZ;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) mine[(32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = o_acc[db][g];
        if (hh == 0) { mine[2048 + r] = m_run; mine[2048 + 32 + r] = l_run; }
    }
    __syncthreads();
    if (kp > 0) return;
#pragma unroll
    for (int part = 1; part < NKP; ++part) {
        const float *oth = mrg + ((part - 1) * NQB + qb) * MSZ;
        const float m_o = oth[2048 + r], l_o = oth[2048 + 32 + r];
        const float m_new = fmaxf(m_run, m_o);
        const float a_me = fast_exp2((m_run - m_new) * c2), a_o = fast_exp2((m_o - m_new) * c2);
        l_run = l_run * a_me + l_o * a_o;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                o_acc[db][g] = o_acc[db][g] * a_me + oth[(32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] * a_o;
    }
    if (CHUNK) {            // un-normalised partial result of this chunk (relpos_attn_fwd_kernel's part format: m in units of the scaled score)
#ifdef AT_PROFILE
        ATS_STAMP(7);
        if (bid == (unsigned)(nqt - 2) && tid == 0) {     // a full-size item of chunk 0 (profile build: its stamps replace real values)
            for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(part_ml)[i] = sacc[i];
            return;
        }
#endif
        if (iq < Tn) {
            const size_t row = (((size_t)b * H + h) * Tn + iq) * nparts + part;
            float *po = part_o + row * 64;
            const float ks = drop ? keep_scale : 1.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(po + 32 * db + 8 * q + 4 * hh) =
                        make_float4(o_acc[db][4 * q] * ks, o_acc[db][4 * q + 1] * ks, o_acc[db][4 * q + 2] * ks, o_acc[db][4 * q + 3] * ks);
            if (hh == 0) *reinterpret_cast<float2 *>(part_ml + row * 2) = make_float2(l_run > 0.f ? m_run * scale : -INFINITY, l_run);
        }
        return;
    }
    if (iq < Tn) {
        const float inv = l_run > 0.f ? (drop ? keep_scale : 1.f) / l_run : 0.f;
        bf16_t *orow = out + ((long long)b * Tn + iq) * D + (long long)h * Dh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * db + 8 * q + 4 * hh;
                st4(orow + d, o_acc[db][4 * q] * inv, o_acc[db][4 * q + 1] * inv, o_acc[db][4 * q + 2] * inv, o_acc[db][4 * q + 3] * inv);
            }
        if (hh == 0 && lse) lse[((long long)b * H + h) * Tn + iq] = m_run * scale + __logf(l_run);
    }
#ifdef AT_PROFILE
    ATS_STAMP(7);   // merge + epilogue
    if (bid == 0 && tid == 0 && lse)
        for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(lse)[i] = sacc[i];   // (profile build: clobbers the first lse values)
#endif
}

template <int QH>
__global__ __launch_bounds__(512, 2) void relpos_attn_fwd_short2_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk,
                                                                        const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                        const int32_t *__restrict__ key_lens, bf16_t *__restrict__ out,
                                                                        float *__restrict__ lse, int Tn, int H, float scale, int causal,
                                                                        float pdrop, unsigned long long seed,
                                                                        const unsigned long long *__restrict__ seed_dev,
                                                                        unsigned short *__restrict__ keepbits) {
    relpos_attn_fwd_short2_item<QH, false>(qkv, pk, bias_u, bias_v, key_lens, out, lse, Tn, H, scale, causal, pdrop, seed, seed_dev, keepbits, blockIdx.x,
                                           gridDim.x, nullptr, nullptr, nullptr);
}
// long sequences: PERSISTENT workgroups (one per CU: the 160 KB of LDS admit no second one) walk the live (query tile, chunk) items - a fresh
// workgroup per item cost ~4 us of launch and teardown around ~10 us of work, and the items beyond the causal limit as much for nothing
__global__ __launch_bounds__(512, 2) void relpos_attn_fwd_chunk_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk,
                                                                       const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                       const int32_t *__restrict__ key_lens, int Tn, int H, float scale, int causal,
                                                                       float pdrop, unsigned long long seed, const unsigned long long *__restrict__ seed_dev,
                                                                       const AttnChunkPlan plan, unsigned nitems, float *__restrict__ part_o,
                                                                       float *__restrict__ part_ml) {
    for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
        relpos_attn_fwd_short2_item<128, true>(qkv, pk, bias_u, bias_v, key_lens, nullptr, nullptr, Tn, H, scale, causal, pdrop, seed, seed_dev, nullptr, item,
                                               nitems, &plan, part_o, part_ml);
        __syncthreads();        // every wave is through the item (the merge buffers live where the next item's tiles land)
    }
}

// =====================================================================================================================
// Backward, query-major pass. Same decomposition as the forward (workgroup = (b, h, QH queries), 8 waves = query blocks x key parts, a lane
// owns one query); recomputes the probabilities from (q, k, p, lse) and leaves, as csrc/attention.hip's relpos_attn_bwd_q_kernel does:
//   dQ = dQ_ac + dQ_bd, per-wave partial sums of d(pos_bias_u) / d(pos_bias_v), and the two [B,H,T,Tp] tensors P_d (dropout applied) and
//   scale * dS that the key-major pass (dK, dV) and the d(pk) pass contract.
// K, V and the band come in by LDS-DMA once (112 KB at T' = 250), everything else is per-lane registers. Per 32 x 32 block of scores:
// 28 MFMAs (S, dP, two G blocks, dQ_ac, dQ_bd) and ~190 vector instructions (round 4's kernel: ~650):
//   p * keep_scale = exp2(fma(x, scale log2 e, log2 keep_scale - lse log2 e)) ; keep factors are AND masks from the forward's keep-bits ;
//   scale * dS = p' * fma(dP & mask, scale, -delta scale / keep_scale) ; G goes through LDS as fp16 pairs (v_fma_mix_f32 adds it) and dG^T
//   comes back as a bf16 tile read with the transposing LDS read (it was an fp32 tile read element by element).
// One tile image serves row reads and transposed reads: chunk swizzle S(row) = ((row >> 1) & 1) << 2 | (row >> 2) & 3.
// =====================================================================================================================
template <int QH>
__global__ __launch_bounds__(512, 2) void relpos_attn_bwd_q_short_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk,
                                                                         const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                         const int32_t *__restrict__ key_lens, const bf16_t *__restrict__ out,
                                                                         const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                                         bf16_t *__restrict__ dqkv, bf16_t *__restrict__ pd_out,
                                                                         bf16_t *__restrict__ ds_out, float *__restrict__ slab_uv, int Tp, int Tn, int H,
                                                                         float scale, int causal, float pdrop, unsigned long long seed,
                                                                         const unsigned long long *__restrict__ seed_dev,
                                                                         const unsigned short *__restrict__ keepbits /*forward's keep-bits or NULL*/) {
    constexpr int Dh = 64, NQB = QH / 32, NKP = 8 / NQB, TPAD = 2 * QH, KP = TPAD / NKP, NSUB = KP / 32, NB = QH + TPAD;
    constexpr int K_OFF = 0, V_OFF = TPAD * 128, P_OFF = 2 * TPAD * 128, G_OFF = P_OFF + NB * 128;
    constexpr int NPC = (2 * TPAD + NB) / 8;                    // DMA pieces of 8 rows: K, V, band
    static_assert(TPAD / 8 % 8 == 0 && NB / 8 % 8 == 0, "every round of 8 pieces has one kind");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    const int nqt = (Tn + QH - 1) / QH, npair = gridDim.x / nqt;
    int qt, pair;
    if ((npair & 7) == 0) {     // the query tiles of one (b, h) on one XCD (speed only)
        const int k = blockIdx.x >> 3;
        qt = k % nqt;
        pair = (k / nqt) * 8 + (blockIdx.x & 7);
    } else {
        qt = blockIdx.x % nqt;
        pair = blockIdx.x / nqt;
    }
    const int b = pair / H, h = pair % H, i0 = qt * QH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int qb = wave % NQB, kp = wave / NQB;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const bf16_t *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const bf16_t *p_base = pk + (long long)h * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int r_lo = (Tn - 1) - (i0 + QH - 1);
    const int iq = i0 + 32 * qb + r, iqc = min(iq, Tn - 1);
    const bool q_ok = iq < Tn;

#ifdef AT_PROFILE
    long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define ATB_STAMP(i) do { const long long n_ = clock64(); acc_t[i] += n_ - t_prev; t_prev = n_; } while (0)
#else
#define ATB_STAMP(i)
#endif
    // ---- this lane's query row: q, dO, O (ordinary loads, in flight beside the DMA pieces), lse, the forward's keep-bits
    float q8[4][8], u8[4][8], v8[4][8], d8[4][8], o8[4][8];
    {
        const bf16_t *qrow = q_base + (long long)iqc * row_stride;
        const bf16_t *orow = out + ((long long)b * Tn + iqc) * D + (long long)h * Dh;
        const bf16_t *dorow = dout + ((long long)b * Tn + iqc) * D + (long long)h * Dh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ld8(qrow + 16 * s + 8 * hh, q8[s]);
            ld8(dorow + 16 * s + 8 * hh, d8[s]);
            ld8(orow + 16 * s + 8 * hh, o8[s]);
            ld8(bias_u + h * Dh + 16 * s + 8 * hh, u8[s]);
            ld8(bias_v + h * Dh + 16 * s + 8 * hh, v8[s]);
        }
    }
    const float lse_i = lse[((long long)b * H + h) * Tn + iqc];
    const bool has_kb = keepbits != nullptr;
    const uint4 kbw = *reinterpret_cast<const uint4 *>(has_kb ? reinterpret_cast<const char *>(keepbits + (((size_t)(b * H + h) * Tn + iqc) * 2 + hh) * 8)
                                                             : reinterpret_cast<const char *>(qkv));
    // ---- K, V, band by LDS-DMA (piece index = wave + 8 n: one kind per n)
    {
        const int prow = lane >> 3, pos = lane & 7;
        const unsigned chunk = (pos ^ ((((prow >> 1) & 1) << 2) | (prow >> 2) | ((wave & 1) << 1))) << 4;      // S(tile row) of this lane's piece row
        const int qs_b = (int)row_stride * 2, ps_b = D * 2;
        const i32x4 srd_q = make_srd(q_base, (unsigned)(Tn * qs_b) - (unsigned)(h * 3 * Dh * 2));
        const i32x4 srd_p = make_srd(p_base, (unsigned)((2 * Tn - 1) * ps_b) - (unsigned)(h * Dh * 2));
#pragma unroll
        for (int n = 0; n < NPC / 8; ++n) {
            const int pc = wave + 8 * n;
            if (8 * n < TPAD / 8)
                dma_piece(srd_q, qs_b, chunk, pc * 8, 0, Tn - 1, Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + K_OFF + pc * 1024), lane);
            else if (8 * n < 2 * TPAD / 8)
                dma_piece(srd_q, qs_b, chunk, (pc - TPAD / 8) * 8, 0, Tn - 1, 2 * Dh * 2, __builtin_amdgcn_readfirstlane(lds0 + V_OFF + (pc - TPAD / 8) * 1024), lane);
            else
                dma_piece(srd_p, ps_b, chunk, r_lo + (pc - 2 * TPAD / 8) * 8, 0, 2 * Tn - 2, 0u,
                          __builtin_amdgcn_readfirstlane(lds0 + P_OFF + (pc - 2 * TPAD / 8) * 1024), lane);
        }
    }
    bf16x8 qu[4], qv[4], dob[4];
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float du8[8], dv8[8], dd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            du8[j] = q8[s][j] + u8[s][j];
            dv8[j] = q8[s][j] + v8[s][j];
            dd[j] = q_ok ? d8[s][j] : 0.f;
            delta += dd[j] * (q_ok ? o8[s][j] : 0.f);
        }
        qu[s] = bf16x8_of(pk_bf16(du8[0], du8[1]), pk_bf16(du8[2], du8[3]), pk_bf16(du8[4], du8[5]), pk_bf16(du8[6], du8[7]));
        qv[s] = bf16x8_of(pk_bf16(dv8[0], dv8[1]), pk_bf16(dv8[2], dv8[3]), pk_bf16(dv8[4], dv8[5]), pk_bf16(dv8[6], dv8[7]));
        dob[s] = bf16x8_of(pk_bf16(dd[0], dd[1]), pk_bf16(dd[2], dd[3]), pk_bf16(dd[4], dd[5]), pk_bf16(dd[6], dd[7]));
    }
    delta += other_half(delta);
    ATB_STAMP(0);   // row loads, DMA issue, operand fragments
    wait_vm_barrier<0>();               // every piece of every wave has landed: the waves run free from here
    ATB_STAMP(1);   // wait for the tiles

    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = pdrop > 0.f ? drop_scale16(thr) : 1.f;
    const bool drop = pdrop > 0.f;
    const unsigned row_state = attn_row_state((unsigned long long)(b * H + h) * Tn + iq, drop_key(seed));
    const float c2 = scale * 1.4426950408889634f;
    const float nb = __log2f(keep_scale) - lse_i * 1.4426950408889634f;        // exp2(fma(x, c2, nb)) = p * keep_scale
    const float dl = delta * scale / keep_scale;
    float one = 1.f;
    asm volatile("" : "+v"(one));
    const int j_max = q_ok ? min(len - 1, causal ? causal_limit(iq, causal) : 0x3fffffff) : -1;     // last key this lane's query attends
    int j_all = j_max;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) j_all = min(j_all, __shfl_xor(j_all, o));
    j_all = __builtin_amdgcn_readfirstlane(j_all);
    const int lim_blk = causal ? causal_limit(min(i0 + 32 * qb + 31, Tn - 1), causal) : 0x3fffffff;
    const int j_first = kp * KP;
    int j_last = min(min((kp + 1) * KP, len), lim_blk + 1);
    if (i0 + 32 * qb >= Tn) j_last = j_first;
    const int nsub = j_last > j_first ? (j_last - j_first + 31) / 32 : 0;      // wave-uniform
    const char *k_lds = smem + K_OFF, *v_lds = smem + V_OFF, *p_lds = smem + P_OFF;
    const unsigned g_addr = lds0 + G_OFF + wave * 4096;                        // this wave's [64 band rows][32 queries] tile: G as fp16, then dG^T as bf16
    const char *g_tile = smem + G_OFF + wave * 4096;
    const int sw_r = (((r >> 1) & 1) << 2) | ((r >> 2) & 3);                   // S(row) of a fragment row 32-aligned + r
    const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    // dG^T fragment masks: element (band row rl = 16 sp + 8 hh + e, query r) of a block exists iff 0 <= rl + r - 31 < 32 (bf16-pair bit masks)
    unsigned dgmask[4][4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl0 = 16 * sp + 8 * hh + 2 * k + r - 31;
            dgmask[sp][k] = ((jl0 >= 0 && jl0 < 32) ? 0xffffu : 0u) | ((jl0 + 1 >= 0 && jl0 + 1 < 32) ? 0xffff0000u : 0u);
        }
    bf16_t *const prow0 = pd_out + (((long long)b * H + h) * Tn + iqc) * Tp + 8 * hh;
    bf16_t *const srow0 = ds_out + (((long long)b * H + h) * Tn + iqc) * Tp + 8 * hh;
    f32x16 dqu[2], dqv[2];
    dqu[0] = dqu[1] = dqv[0] = dqv[1] = (f32x16){0};

#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
        const int jb = j_first + 32 * sub;
        const int Rb = jb - 32 * qb + QH - 32;      // band rows Rb .. Rb + 63 of the workgroup's band tile
        f32x16 s_acc = {0}, dpd = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(k_lds + (jb + r) * 128 + (((2 * s + hh) ^ sw_r) << 4));
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qu[s], s_acc, 0, 0, 0);
            const bf16x8 va = *reinterpret_cast<const bf16x8 *>(v_lds + (jb + r) * 128 + (((2 * s + hh) ^ sw_r) << 4));
            dpd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dob[s], dpd, 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x16 g_acc = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(p_lds + (Rb + 32 * rb + r) * 128 + (((2 * s + hh) ^ sw_r) << 4));
                g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qv[s], g_acc, 0, 0, 0);
            }
            const unsigned a = g_addr + (32 * rb + 4 * hh) * 64 + 2 * r;
#define AT2_GSTORE(G) g_store_pair<(((G) & 3) + 8 * ((G) >> 2)) * 64>(a, pk_f16(g_acc[G], g_acc[(G) + 1]), false)
            AT2_GSTORE(0); AT2_GSTORE(2); AT2_GSTORE(4); AT2_GSTORE(6); AT2_GSTORE(8); AT2_GSTORE(10); AT2_GSTORE(12); AT2_GSTORE(14);
#undef AT2_GSTORE
        }
        ATB_STAMP(2);   // S, dP, G MFMAs + G tile to LDS
        const unsigned a_skew = g_addr + (31 - r + 4 * hh) * 64 + 2 * r;
        unsigned bd[16];
        asm volatile(
            "ds_read_u16 %0, %16 offset:0\n\tds_read_u16 %1, %16 offset:64\n\t"
            "ds_read_u16 %2, %16 offset:128\n\tds_read_u16 %3, %16 offset:192\n\t"
            "ds_read_u16 %4, %16 offset:512\n\tds_read_u16 %5, %16 offset:576\n\t"
            "ds_read_u16 %6, %16 offset:640\n\tds_read_u16 %7, %16 offset:704\n\t"
            "ds_read_u16 %8, %16 offset:1024\n\tds_read_u16 %9, %16 offset:1088\n\t"
            "ds_read_u16 %10, %16 offset:1152\n\tds_read_u16 %11, %16 offset:1216\n\t"
            "ds_read_u16 %12, %16 offset:1536\n\tds_read_u16 %13, %16 offset:1600\n\t"
            "ds_read_u16 %14, %16 offset:1664\n\tds_read_u16 %15, %16 offset:1728\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(bd[0]), "=&v"(bd[1]), "=&v"(bd[2]), "=&v"(bd[3]), "=&v"(bd[4]), "=&v"(bd[5]), "=&v"(bd[6]), "=&v"(bd[7]),
              "=&v"(bd[8]), "=&v"(bd[9]), "=&v"(bd[10]), "=&v"(bd[11]), "=&v"(bd[12]), "=&v"(bd[13]), "=&v"(bd[14]), "=&v"(bd[15])
            : "v"(a_skew)
            : "memory");
        unsigned kw = 0xffffu;   // keep-bits of this lane's 16 keys (bit g = accumulator element g)
        if (drop) {
            if (has_kb) {       // workgroup-uniform; jb >> 5 is wave-uniform
                const int sb = jb >> 5;
                const unsigned pr = (sb >> 1) == 0 ? kbw.x : (sb >> 1) == 1 ? kbw.y : (sb >> 1) == 2 ? kbw.z : kbw.w;
                kw = (sb & 1) ? (pr >> 16) : (pr & 0xffffu);
            } else {
                kw = attn_keep16(row_state, jb >> 5, hh, thr);
            }
        }
        float pp[16], ds[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float t = add_h_lo(s_acc[g], bd[g], one);
            pp[g] = fast_exp2(__builtin_fmaf(t, c2, nb));
        }
        if (jb + 31 > j_all) {      // (wave-uniform) key padding, look-ahead mask, rows beyond the sequence
#pragma unroll
            for (int g = 0; g < 16; ++g) pp[g] = (jb + (g & 3) + 8 * (g >> 2) + 4 * hh > j_max) ? 0.f : pp[g];
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const unsigned e = (unsigned)((int)(kw << (31 - g)) >> 31);       // all ones iff kept (v_bfe_i32)
            const float tg = __uint_as_float(__float_as_uint(dpd[g]) & e);
            ds[g] = pp[g] * __builtin_fmaf(tg, scale, -dl);
            pp[g] = __uint_as_float(__float_as_uint(pp[g]) & e);
        }
        unsigned pdw[8], dsw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pdw[k] = pk_bf16(pp[2 * k], pp[2 * k + 1]);
            dsw[k] = pk_bf16(ds[2 * k], ds[2 * k + 1]);
        }
        ATB_STAMP(3);   // skewed read, p, dS
        // P_d and scale*dS rows: the two half-waves of a query hold alternate runs of four keys; one v_permlane32_swap per register pairs
        // them up into runs of eight (16-byte stores: half the store instructions; the store tail was 15 % of the kernel)
        {
            unsigned po[8], so[8];
#pragma unroll
            for (int q = 0; q < 4; q += 2)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const auto rp = __builtin_amdgcn_permlane32_swap(pdw[2 * q + e], pdw[2 * q + 2 + e], false, false);
                    po[2 * q + e] = rp[0]; po[2 * q + 2 + e] = rp[1];
                    const auto rs = __builtin_amdgcn_permlane32_swap(dsw[2 * q + e], dsw[2 * q + 2 + e], false, false);
                    so[2 * q + e] = rs[0]; so[2 * q + 2 + e] = rs[1];
                }
            if (q_ok) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {    // lanes 0-31: keys 8q .. 8q+7 ; lanes 32-63: keys 8(q+1) .. 8(q+1)+7
                    *reinterpret_cast<uint4 *>(prow0 + jb + 8 * q) = make_uint4(po[2 * q], po[2 * q + 1], po[2 * q + 2], po[2 * q + 3]);
                    *reinterpret_cast<uint4 *>(srow0 + jb + 8 * q) = make_uint4(so[2 * q], so[2 * q + 1], so[2 * q + 2], so[2 * q + 3]);
                }
            }
        }
        // inverse skew: dG^T[jl - i + 31][i] = scale dS[i, jl], as bf16 over the G tile (the skewed read above has completed)
#define AT2_DSTORE(K) g_store_pair<((2 * (K) & 3) + 8 * (2 * (K) >> 2)) * 64>(a_skew, dsw[K], false)
        AT2_DSTORE(0); AT2_DSTORE(1); AT2_DSTORE(2); AT2_DSTORE(3); AT2_DSTORE(4); AT2_DSTORE(5); AT2_DSTORE(6); AT2_DSTORE(7);
#undef AT2_DSTORE
        ATB_STAMP(4);   // P_d / dS stores, inverse skew
        const bf16x8 dsb0 = bf16x8_of(dsw[0], dsw[1], dsw[2], dsw[3]), dsb1 = bf16x8_of(dsw[4], dsw[5], dsw[6], dsw[7]);
        // dQ_ac^T += K^T . dSs^T   (A = K^T through the transposing read; k order of dsb = accumulator row order)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row_lo = jb + 16 * s + 4 * hh + q4, row_hi = row_lo + 8, col = 32 * db + 16 * mhalf + 4 * p4;
                const int s_lo = (((row_lo >> 1) & 1) << 2) | ((row_lo >> 2) & 3), s_hi = (((row_hi >> 1) & 1) << 2) | ((row_hi >> 2) & 3);
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(k_lds + row_lo * 128 + (((col >> 3) ^ s_lo) << 4) + (col & 7) * 2));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(k_lds + row_hi * 128 + (((col >> 3) ^ s_hi) << 4) + (col & 7) * 2));
                bf16x8 ka;
                ka[0] = lo[0]; ka[1] = lo[1]; ka[2] = lo[2]; ka[3] = lo[3]; ka[4] = hi[0]; ka[5] = hi[1]; ka[6] = hi[2]; ka[7] = hi[3];
                dqu[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, s == 0 ? dsb0 : dsb1, dqu[db], 0, 0, 0);
            }
        // dQ_bd^T += Pband^T . dG^T   (natural k order: band rows 16 sp + 8 hh + e; both operands through the transposing read)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the dG stores above (asm) are in the tile before the builtin reads below are issued (in order anyway)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const int grow = 16 * sp + 8 * hh + q4, gcol = 16 * mhalf + 4 * p4;
            const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(g_tile + grow * 64 + gcol * 2));
            const bf16x4 ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(g_tile + (grow + 4) * 64 + gcol * 2));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 gl = __builtin_bit_cast(u32x2, glo), gh = __builtin_bit_cast(u32x2, ghi);
            const bf16x8 dgb = bf16x8_of(gl[0] & dgmask[sp][0], gl[1] & dgmask[sp][1], gh[0] & dgmask[sp][2], gh[1] & dgmask[sp][3]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row_lo = Rb + 16 * sp + 8 * hh + q4, row_hi = row_lo + 4, col = 32 * db + 16 * mhalf + 4 * p4;
                const int s_lo = (((row_lo >> 1) & 1) << 2) | ((row_lo >> 2) & 3), s_hi = (((row_hi >> 1) & 1) << 2) | ((row_hi >> 2) & 3);
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p_lds + row_lo * 128 + (((col >> 3) ^ s_lo) << 4) + (col & 7) * 2));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p_lds + row_hi * 128 + (((col >> 3) ^ s_hi) << 4) + (col & 7) * 2));
                bf16x8 pa;
                pa[0] = lo[0]; pa[1] = lo[1]; pa[2] = lo[2]; pa[3] = lo[3]; pa[4] = hi[0]; pa[5] = hi[1]; pa[6] = hi[2]; pa[7] = hi[3];
                dqv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, dgb, dqv[db], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile's reads have returned before the next block's G stores (asm) overwrite it
        ATB_STAMP(5);   // dQ MFMAs
    }
    // ---- the key parts of a query block: parts kp > 0 hand their dQ shares to part 0 through LDS (K / V / band are dead now), one part per round
    float *xch = reinterpret_cast<float *>(smem) + qb * 64 * 64;      // [qb][register][lane]: conflict-free
    static_assert(NQB * 64 * 64 * 4 <= G_OFF, "exchange buffers fit the dead tiles");
#pragma unroll 1
    for (int part = 1; part < NKP; ++part) {
        __syncthreads();
        if (kp == part) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    xch[(db * 16 + g) * 64 + lane] = dqu[db][g];
                    xch[(32 + db * 16 + g) * 64 + lane] = dqv[db][g];
                }
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    dqu[db][g] += xch[(db * 16 + g) * 64 + lane];
                    dqv[db][g] += xch[(32 + db * 16 + g) * 64 + lane];
                }
        }
    }
    __syncthreads();
    ATB_STAMP(6);   // merge of the key parts
    if (kp > 0) return;
    if (q_ok) {
        bf16_t *dq = dqkv + ((long long)b * Tn + iq) * row_stride + (long long)h * 3 * Dh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; g += 4)
                st4(dq + 32 * db + 8 * (g >> 2) + 4 * hh, dqu[db][g] + dqv[db][g], dqu[db][g + 1] + dqv[db][g + 1], dqu[db][g + 2] + dqv[db][g + 2],
                    dqu[db][g + 3] + dqv[db][g + 3]);
    }
    // ---- partial sums over this wave's 32 queries for d(pos_bias_u) (= sum dQ_ac) and d(pos_bias_v) (= sum dQ_bd): the accumulators go through
    // LDS as [dim][query] and lane d adds up row d, starting at column d (the lanes of a read fall into different banks). Rows beyond the
    // sequence hold exact zeros (j_max = -1).
    float *scr = reinterpret_cast<float *>(smem + NQB * 64 * 64 * 4) + qb * 64 * 32;      // behind the exchange buffers
    static_assert(NQB * 64 * 64 * 4 + NQB * 64 * 32 * 4 <= G_OFF + 8 * 4096, "column-sum scratch fits");
    float *slab = slab_uv + (((long long)b * (4 * ((Tn + 127) / 128)) + (i0 >> 5) + qb) * H + h) * 128;   // part = (b, block of 32 queries); row = [h][u 64 | v 64]
    float colsum[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) scr[(32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = which ? dqv[db][g] : dqu[db][g];
        __builtin_amdgcn_wave_barrier();
        float part_sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 32; ++k) part_sum[k & 3] += scr[lane * 32 + ((k + lane) & 31)];
        colsum[which] = (part_sum[0] + part_sum[1]) + (part_sum[2] + part_sum[3]);
    }
    slab[lane] = colsum[0];
    slab[64 + lane] = colsum[1];
    if (QH == 64 && Tn <= 64) {     // one workgroup per (b, h): the slab rows of the two query blocks that do not exist must read as zero
        slab[2 * H * 128 + lane] = 0.f;
        slab[2 * H * 128 + 64 + lane] = 0.f;
    }
#ifdef AT_PROFILE
    ATB_STAMP(7);   // dQ store, bias partial sums
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(slab)[i] = acc_t[i];     // (profile build: over this wave's own partial sums)
#endif
}

// =====================================================================================================================
// Backward, key-major pass on the materialised P_d / scale*dS (the contraction of csrc/attention.hip's relpos_attn_bwd_kv2_kernel):
//   dV^T[d][j] = sum_i dO[i][d] P_d[i][j] ,   dK^T[d][j] = sum_i (q_i + u)[d] dS[i][j] = sum_i q_i[d] dS[i][j] + u[d] sum_i dS[i][j]
// workgroup = (b, h, 128 keys), 8 waves = (32 keys) x (32 head dims); chunks of 64 queries stream through a three-stage LDS ring by LDS-DMA
// (P_d and dS rows of 256 bytes, dO and Q rows of 128), every operand fragment comes out of it through the transposing read (k = query
// index = tile row), and the pos_bias_u term is one more MFMA per k-step against a constant fragment - the kernel has no vector
// arithmetic at all in its loop (round 4's kernel staged the tiles through registers with 8 fp32 adds + conversions per piece and
// re-read dO / Q once per 64 keys: 16.9 us per layer; the tiles' bytes are the floor here).
// Also leaves the (q + pos_bias_v) rows in the [H, B*T, Dh] layout of the d(pk) pass (workgroups of the first key half).
// =====================================================================================================================
__global__ __launch_bounds__(512, 2) void relpos_attn_bwd_kv_short_kernel(const bf16_t *__restrict__ qkv, const float *__restrict__ bias_u,
                                                                          const float *__restrict__ bias_v, const int32_t *__restrict__ key_lens,
                                                                          const bf16_t *__restrict__ dout, const bf16_t *__restrict__ pd,
                                                                          const bf16_t *__restrict__ ds, bf16_t *__restrict__ dqkv,
                                                                          bf16_t *__restrict__ qv_out /*[H][B*T][Dh]*/, int Bn, int Tn, int Tp, int H, int causal) {
    constexpr int Dh = 64, STAGE = 48 * 1024, PD_OFF = 0, DS_OFF = 16 * 1024, DO_OFF = 32 * 1024, Q_OFF = 40 * 1024, PW = 6;   // PW: pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkh = (Tn + 127) / 128, npair = gridDim.x / nkh;
    int kh, pair;
    if ((npair & 7) == 0) {     // the key halves of one (b, h) on one XCD: they read the same dO / Q rows (speed only)
        const int k = blockIdx.x >> 3;
        kh = k % nkh;
        pair = (k / nkh) * 8 + (blockIdx.x & 7);
    } else {
        kh = blockIdx.x % nkh;
        pair = blockIdx.x / nkh;
    }
    const int b = pair / H, h = pair % H, j0 = kh * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int kb = wave & 3, dh = wave >> 2;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    // queries that can reach these keys: all of them, or (look-ahead mask) those from the chunk of the first key on
    const int c_first = causal ? (((j0 / causal) * causal) / 64) : 0, c_end = (Tn + 63) / 64;      // chunks [c_first, c_end)

    const bf16_t *pd_base = pd + (((long long)b * H + h) * Tn) * Tp + j0, *ds_base = ds + (((long long)b * H + h) * Tn) * Tp + j0;
    const bf16_t *do_base = dout + ((long long)b * Tn) * D + (long long)h * Dh;
    const bf16_t *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const i32x4 srd_pd = make_srd(pd_base, (unsigned)(Tn * Tp * 2) - (unsigned)(j0 * 2));
    const i32x4 srd_ds = make_srd(ds_base, (unsigned)(Tn * Tp * 2) - (unsigned)(j0 * 2));
    const i32x4 srd_do = make_srd(do_base, (unsigned)(Tn * D * 2) - (unsigned)(h * Dh * 2));
    const i32x4 srd_q = make_srd(q_base, (unsigned)(Tn * (int)row_stride * 2) - (unsigned)(h * 3 * Dh * 2));
    // lane constants of the DMA pieces. P_d / dS tile: rows of 256 bytes, piece = 4 rows, slot (row, ch) <- source chunk ch ^ ((row & 3) << 2 | (row >> 2) & 3);
    // dO / Q tile: rows of 128 bytes, piece = 8 rows, slot (row, pos) <- source chunk pos ^ S(row)
    const int row4 = lane >> 4, ch16 = lane & 15, prow = lane >> 3, pos = lane & 7;
    const unsigned chunk_w = (ch16 ^ (row4 << 2) ^ (wave & 3)) << 4;
    const unsigned chunk_n = (pos ^ ((((prow >> 1) & 1) << 2) | (prow >> 2) | ((wave & 1) << 1))) << 4;
    auto issue_chunk = [&](int c) {     // chunk c -> stage (c - c_first) % 3
        const unsigned st = lds0 + (unsigned)((c - c_first) % 3) * STAGE;
        const int i0c = c * 64;
#pragma unroll
        for (int n = 0; n < PW; ++n) {
            const int pc = wave + 8 * n;
            if (n < 2 || n < 4) {
                const int pq = pc & 15, srow = i0c + pq * 4;
                const int rel = min(max(row4, -srow), Tn - 1 - srow);
                const unsigned voff = __umul24((unsigned)rel & 0xffffffu, (unsigned)(Tp * 2)) + chunk_w + (unsigned)(srow * Tp * 2);
                dma_buf16(n < 2 ? srd_pd : srd_ds, voff, 0u, __builtin_amdgcn_readfirstlane(st + (n < 2 ? PD_OFF : DS_OFF) + pq * 1024));
            } else if (n == 4)
                dma_piece(srd_do, D * 2, chunk_n, i0c + (pc - 32) * 8, 0, Tn - 1, 0u, __builtin_amdgcn_readfirstlane(st + DO_OFF + (pc - 32) * 1024), lane);
            else
                dma_piece(srd_q, (int)row_stride * 2, chunk_n, i0c + (pc - 40) * 8, 0, Tn - 1, 0u, __builtin_amdgcn_readfirstlane(st + Q_OFF + (pc - 40) * 1024), lane);
        }
    };
    if (c_first < c_end) issue_chunk(c_first);
    if (c_first + 1 < c_end) issue_chunk(c_first + 1);

    const int grp = lane >> 4, gh = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    bf16x8 a_u;     // pos_bias_u[d] in every k of row d
    {
        const bf16_t ub = (bf16_t)bias_u[h * Dh + 32 * dh + r];
#pragma unroll
        for (int j = 0; j < 8; ++j) a_u[j] = ub;
    }
    float bv8[8];   // pos_bias_v pieces of the (q + v) rows this thread writes
    ld8(bias_v + h * Dh + (tid & 7) * 8, bv8);
    f32x16 dk = {0}, dv = {0};
    for (int c = c_first; c < c_end; ++c) {
        // chunk c has landed (own pieces: all but chunk c + 1's; then every wave's); the stage of chunk c - 1 is free for chunk c + 2
        if (c + 1 < c_end) wait_vm_barrier<PW>();
        else wait_vm_barrier<0>();
        if (c + 2 < c_end) issue_chunk(c + 2);
        char *st = smem + ((c - c_first) % 3) * STAGE;
        const int i0c = c * 64;
        if (i0c + 64 > Tn) {        // (last chunk) rows beyond the sequence hold copies of the last row: they must add nothing
            const int z0 = Tn - i0c;            // first dead row
            for (int e = tid; e < (64 - z0) * 48; e += 512) {       // 48 16-byte slots per row over the four tiles
                const int rr = z0 + e / 48, sl = e % 48;
                char *dst = sl < 16 ? st + PD_OFF + rr * 256 + sl * 16 : sl < 32 ? st + DS_OFF + rr * 256 + (sl - 16) * 16
                          : sl < 40 ? st + DO_OFF + rr * 128 + (sl - 32) * 16 : st + Q_OFF + rr * 128 + (sl - 40) * 16;
                *reinterpret_cast<uint4 *>(dst) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();        // (no DMA of this wave is in flight here: the last chunk's wait was vmcnt(0))
        }
        if (kh == 0 && qv_out) {    // (q + pos_bias_v) rows of this chunk for the d(pk) pass: thread = (row, 8 dims)
            const int rr = tid >> 3, cc = tid & 7, i = i0c + rr;
            const int sw = (((rr >> 1) & 1) << 2) | ((rr >> 2) & 3);
            const uint4 w = *reinterpret_cast<const uint4 *>(st + Q_OFF + rr * 128 + ((cc ^ sw) << 4));
            if (i < Tn) {
                float f[8];
                f[0] = __uint_as_float(w.x << 16) + bv8[0]; f[1] = __uint_as_float(w.x & 0xffff0000u) + bv8[1];
                f[2] = __uint_as_float(w.y << 16) + bv8[2]; f[3] = __uint_as_float(w.y & 0xffff0000u) + bv8[3];
                f[4] = __uint_as_float(w.z << 16) + bv8[4]; f[5] = __uint_as_float(w.z & 0xffff0000u) + bv8[5];
                f[6] = __uint_as_float(w.w << 16) + bv8[6]; f[7] = __uint_as_float(w.w & 0xffff0000u) + bv8[7];
                st8(qv_out + (((long long)h * Bn + b) * Tn + i) * Dh + cc * 8, f);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // 16 queries per MFMA step
            const int row_lo = 16 * s + 8 * hh + q4, row_hi = row_lo + 4;
            // B fragments: P_d / dS [k = query][col = key 32 kb + ...]
            const int chw = 4 * kb + 2 * gh + (p4 >> 1), inner = (p4 & 1) * 8;
            const int sw_lo = ((row_lo & 3) << 2) | ((row_lo >> 2) & 3), sw_hi = ((row_hi & 3) << 2) | ((row_hi >> 2) & 3);
            const int o_lo = row_lo * 256 + ((chw ^ sw_lo) << 4) + inner, o_hi = row_hi * 256 + ((chw ^ sw_hi) << 4) + inner;
            // A fragments: dO^T / Q^T [row = dim 32 dh + ...][k = query]
            const int chn = 4 * dh + 2 * gh + (p4 >> 1);
            const int sn_lo = (((row_lo >> 1) & 1) << 2) | ((row_lo >> 2) & 3), sn_hi = (((row_hi >> 1) & 1) << 2) | ((row_hi >> 2) & 3);
            const int n_lo = row_lo * 128 + ((chn ^ sn_lo) << 4) + inner, n_hi = row_hi * 128 + ((chn ^ sn_hi) << 4) + inner;
            auto frag = [&](const char *tile, int lo_off, int hi_off) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + lo_off));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + hi_off));
                bf16x8 f;
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                return f;
            };
            const bf16x8 b_p = frag(st + PD_OFF, o_lo, o_hi), b_s = frag(st + DS_OFF, o_lo, o_hi);
            const bf16x8 a_do = frag(st + DO_OFF, n_lo, n_hi), a_q = frag(st + Q_OFF, n_lo, n_hi);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, b_p, dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_q, b_s, dk, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_u, b_s, dk, 0, 0, 0);
        }
    }
    // accumulators: rows = head dims 32 dh + (g&3) + 8(g>>2) + 4hh, column = this lane's key; masked keys get zeros
    const int jk = j0 + 32 * kb + r;
    if (jk < Tn) {
        const bool k_live = jk < len;
        bf16_t *dkp = dqkv + ((long long)b * Tn + jk) * row_stride + (long long)h * 3 * Dh + Dh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * dh + 8 * q + 4 * hh;
            float kk[4], vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { kk[e] = k_live ? dk[4 * q + e] : 0.f; vv[e] = k_live ? dv[4 * q + e] : 0.f; }
            st4(dkp + d, kk[0], kk[1], kk[2], kk[3]);
            st4(dkp + Dh + d, vv[0], vv[1], vv[2], vv[3]);
        }
    }
}

extern "C" {

/* The short-sequence forward (bf16, Dh = 64, 2 <= T <= 256): called by tsasr_relpos_attn_fwd_ws (csrc/attention.hip). */
int tsasr_attn_short_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, void *out, float *lse,
                         int B, int T, int H, float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev,
                         void *keepbits, hipStream_t st) {
    if (T > 128) {
        constexpr int LDSS = (2 * 256 + 128 + 256) * 128 + 8 * 6144;   // K, V (256 rows), band (384 rows), 8 x 3 fp16 G slots
        (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_short2_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_fwd_short2_kernel<128><<<dim3(cdiv(T, 128) * H * B), 512, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens,
                                                                                      (bf16_t *)out, lse, T, H, scale, causal, pdrop, seed, seed_dev,
                                                                                      (unsigned short *)keepbits);
    } else {
        constexpr int LDSS = (2 * 128 + 64 + 128) * 128 + 8 * 6144;
        (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_short2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_fwd_short2_kernel<64><<<dim3(cdiv(T, 64) * H * B), 512, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens,
                                                                                    (bf16_t *)out, lse, T, H, scale, causal, pdrop, seed, seed_dev,
                                                                                    (unsigned short *)keepbits);
    }
    return 0;
}

/* The same kernel per chunk of 256 keys for long sequences (bf16, Dh = 64, T > 256): partial results for relpos_attn_merge_kernel,
 * nparts = cdiv(T, 256) parts per query (part_o [B*H*T*nparts][64], part_ml [B*H*T*nparts][2], fp32). */
int tsasr_attn_chunk_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, int B, int T, int H,
                         float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int nparts,
                         float *part_o, float *part_ml, hipStream_t st) {
    constexpr int LDSS = (2 * 256 + 128 + 256) * 128 + 8 * 6144;
    if (nparts > 64) return 1;      // (T > 16384: the caller keeps the streaming kernel)
    AttnChunkPlan plan{};
    const int nqt = cdiv(T, 128);
    plan.nparts = nparts;
    for (int p = 0; p < nparts; ++p) {
        int first = nqt;
        for (int qt = 0; qt < nqt; ++qt) {
            const int i_last = std::min(qt * 128 + 127, T - 1 + 128);
            const int lim = causal <= 1 ? i_last : (i_last / causal + 1) * causal - 1;      // = causal_limit (csrc/attn_common.h)
            const int j_end = causal ? std::min(T, lim + 1) : T;
            if (256 * p < j_end) { first = qt; break; }
        }
        plan.first_qt[p] = first;
        plan.base[p + 1] = plan.base[p] + (nqt - first);
    }
    plan.per_pair = plan.base[nparts];
    const unsigned nitems = (unsigned)plan.per_pair * (unsigned)(B * H);
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
    relpos_attn_fwd_chunk_kernel<<<dim3(std::min<unsigned>(nitems, (unsigned)cus)), 512, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, T, H,
                                                                                                  scale, causal, pdrop, seed, seed_dev, plan, nitems, part_o, part_ml);
    return 0;
}

/* The short-sequence query-major backward pass (bf16, Dh = 64, 2 <= T <= 256): called by tsasr_relpos_attn_bwd (csrc/attention.hip) in place of
 * relpos_attn_bwd_q_kernel; same outputs and workspace layout (slab rows: one per block of 32 queries). */
int tsasr_attn_short_bwd_q(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *out,
                           const void *dout, const float *lse, void *dqkv, void *pd, void *ds, float *slab, int Tp, int B, int T, int H, float scale,
                           int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, const void *keepbits, hipStream_t st) {
    if (T > 128) {
        constexpr int LDSS = (2 * 256 + 128 + 256) * 128 + 8 * 4096;
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_q_short_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_bwd_q_short_kernel<128><<<dim3(cdiv(T, 128) * H * B), 512, LDSS, st>>>(
            (const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, (bf16_t *)pd,
            (bf16_t *)ds, slab, Tp, T, H, scale, causal, pdrop, seed, seed_dev, (const unsigned short *)keepbits);
    } else {
        constexpr int LDSS = (2 * 128 + 64 + 128) * 128 + 8 * 4096;
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_q_short_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        relpos_attn_bwd_q_short_kernel<64><<<dim3(cdiv(T, 64) * H * B), 512, LDSS, st>>>(
            (const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, (bf16_t *)pd,
            (bf16_t *)ds, slab, Tp, T, H, scale, causal, pdrop, seed, seed_dev, (const unsigned short *)keepbits);
    }
    return 0;
}

/* The short-sequence key-major backward pass (bf16, Dh = 64, 2 <= T <= 256): called by tsasr_relpos_attn_bwd in place of relpos_attn_bwd_kv2_kernel. */
int tsasr_attn_short_bwd_kv(const void *qkv, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *dout, const void *pd,
                            const void *ds, void *dqkv, void *qv_out, int B, int T, int Tp, int H, int causal, hipStream_t st) {
    constexpr int LDSS = 3 * 48 * 1024;
    (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_kv_short_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
    relpos_attn_bwd_kv_short_kernel<<<dim3(cdiv(T, 128) * H * B), 512, LDSS, st>>>((const bf16_t *)qkv, bias_u, bias_v, key_lens, (const bf16_t *)dout,
                                                                                (const bf16_t *)pd, (const bf16_t *)ds, (bf16_t *)dqkv, (bf16_t *)qv_out, B, T,
                                                                                Tp, H, causal);
    return 0;
}

}  // extern "C"
