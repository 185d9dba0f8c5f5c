// Fused relative-position multi-head self-attention (Transformer-XL style, as speechbrain's RelPosMHAXL) for gfx950.
//
// Replaces the body of RelPosMHAXL.forward, vendor/speechbrain/speechbrain/nnet/attention.py:485-639, between the input
// projection and the output projection: q+u / q+v (:586-592), matrix_ac (:595), matrix_bd + rel_shift (:597-598, 468-483),
// scale by 1/sqrt(embed_dim) (:604), look-ahead / key-padding masks filled with -inf (:607-623), softmax, dropout, .V (:625-633).
// The reference materialises AC [B,H,T,T], BDraw [B,H,T,2T-1], the shifted copy, the masked copy, the softmax and the
// dropped copy in HBM (>= 8 passes over 96 MB per layer at B=32, T=250). Here nothing of size T x T leaves the chip.
//
//   score[i,j] = scale * ( (q_i + u) . k_j  +  (q_i + v) . p_{j - i + T - 1} )        (rel_shift closed form)
//   out[i]     = sum_j dropout(softmax_j(score[i,:]))[j] * v_j
//
// Mapping (flash-style, one pass over the keys, online softmax in fp32):
//   * workgroup = (b, h, 128 consecutive queries), 4 waves x 32 queries; key tiles of 64 stream through LDS together with
//     the band of positional rows p_r they can touch (191 rows).
//   * every product runs on v_mfma_f32_32x32x16_bf16 in the TRANSPOSED orientation (rows = keys / band rows / head dims in
//     the accumulator registers, column = query on the lane): a lane owns ONE query, so the softmax row statistics are
//     in-lane reductions plus one cross-half exchange, and the probability tile is directly the B operand of P.V
//     (guide section 3 "accumulator tile as the next MFMA's operand").
//   * the relative shift BD[i,j] = G[i, j-i+31] (G = (Q+v).Pband^T for a 32x32 block) is a per-lane diagonal read: G^T goes
//     through a per-wave fp32 LDS tile [64 r][32 i]; with row stride 32 the skewed read (r = j - i + 31, column i) hits bank
//     (i mod 32): conflict-free.
//   * V is consumed through ds_read_b64_tr_b16 (hardware transpose read) so that it can be staged row-major/coalesced.
// qkv layout: [B, T, H, 3*Dh] with Q|K|V interleaved per head (attention.py:549-553); pk: [2T-1, H*Dh];
// pos_bias_u/v: the (Dh, H) parameter's storage reinterpreted as [H, Dh] (a view, not a transpose; attention.py:586-592).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "attn_common.h"

#ifndef AT_NW
#define AT_NW 4           // waves per workgroup (forward and query-major backward kernels)
#endif
#define AT_TH (64 * AT_NW) // threads per workgroup
#define AT_QW 32          // queries per wave
#define AT_QB (32 * AT_NW) // queries per workgroup
#define AT_KT 64          // keys per LDS tile
#define AT_DP 64          // padded head dim (Dh <= 64)
#define AT_LD 72          // LDS row stride in bf16 (144 B: 16-byte slots rotate by 9 per row -> conflict-free b128 reads)
#define AT_BAND (AT_QB + AT_KT)  // 192 band rows per tile (191 used)

// Loads in these kernels are ALWAYS issued (clamped addresses) and masked afterwards: a conditional load lives in its own basic
// block and is waited for where it stands, so a prologue of 160 guarded element loads or a staging loop of six guarded row
// pieces costs that many serialized memory round trips (the same finding as in csrc/rnnt.hip's backward kernels).

// sum over the 32 lanes of this lane's half of the wave (hh = lane >> 5): DPP inside the 16-lane rows, two readlanes per half
__device__ __forceinline__ float half_sum(float v, int hh) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    const float lo = lane_bcast(v, 0) + lane_bcast(v, 16), hi = lane_bcast(v, 32) + lane_bcast(v, 48);
    return hh ? hi : lo;
}

// 8 consecutive elements row[d0 .. d0+8), zero at and beyond Dh. fast: Dh % 8 == 0 (16-byte aligned pieces)
template <typename T>
__device__ __forceinline__ void load8_clamped(const T *__restrict__ row, int d0, int Dh, bool fast, float (&v)[8]) {
    if (fast) {
        ld8(row + min(d0, Dh - 8), v);
        if (d0 >= Dh) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = ld1(row + min(d0 + j, Dh - 1));
            v[j] = (d0 + j < Dh) ? x : 0.f;
        }
    }
}

// The same 8 elements WITHOUT the zeroing of the dims at and beyond Dh (the address is clamped into the row; the caller masks when it
// consumes the values): load8_clamped's mask overwrites the load's destination, i.e. waits for the load on the spot - a sequence of
// calls was a sequence of round trips (20 of them at the head of relpos_attn_bwd_q). FAST as a compile-time flag: no branch per call.
template <typename T, bool FAST>
__device__ __forceinline__ void load8_raw(const T *__restrict__ row, int d0, int Dh, float (&v)[8]) {
    if constexpr (FAST) ld8(row + min(d0, Dh - 8), v);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ld1(row + min(d0 + j, Dh - 1));
    }
}

// stage ROWS rows of Dh elements (row r -> src + r*src_stride, or zeros when r is outside [lo,hi)) into lds[ROWS][AT_LD]:
// all of a thread's pieces are requested before the first one is stored
template <typename T, int ROWS>
__device__ __forceinline__ void stage_rows(bf16_t *lds, const T *__restrict__ src, long long src_stride, int first_row, int lo, int hi,
                                           int Dh) {
    constexpr int NIT = ROWS * (AT_DP / 8) / AT_TH;
    static_assert(ROWS * (AT_DP / 8) % AT_TH == 0, "whole passes of the workgroup");
    const bool fast = (Dh % 8) == 0;
    const int c = ((threadIdx.x & (AT_TH - 1)) % (AT_DP / 8)) * 8;      // 256 % (AT_DP / 8) == 0: one column group per thread
    float v[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = ((threadIdx.x & (AT_TH - 1)) + it * AT_TH) / (AT_DP / 8), r = first_row + rr;
        load8_clamped<T>(src + (long long)min(max(r, lo), hi - 1) * src_stride, c, Dh, fast, v[it]);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = ((threadIdx.x & (AT_TH - 1)) + it * AT_TH) / (AT_DP / 8), r = first_row + rr;
        if (r < lo || r >= hi) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
        }
        st8(lds + rr * AT_LD + c, v[it]);
    }
}

// The same staging split in two halves so that a tile can be REQUESTED one iteration ahead and written to LDS when the previous
// tile has been consumed: request() issues the loads (clamped, unconditional) into registers in the storage type, commit() masks
// and stores them. Between the two sits the MFMA work of the current tile - the key/query-tile loops no longer expose one memory
// round trip per tile (4 tiles at T' = 250).
template <typename T, int ROWS>
struct StagePieces {
    static constexpr int NIT = ROWS * (AT_DP / 8) / AT_TH, W = 8 * (int)sizeof(T) / 16;   // uint4 words per 8-element piece
    uint4 raw[NIT][W];
    int first_row, lo, hi;
    __device__ __forceinline__ void request(const T *__restrict__ src, long long src_stride, int first, int lo_, int hi_, int Dh) {
        first_row = first; lo = lo_; hi = hi_;
        const int c = ((threadIdx.x & (AT_TH - 1)) % (AT_DP / 8)) * 8, cc = min(c, ((Dh + 7) & ~7) - 8);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = first + (int)((threadIdx.x & (AT_TH - 1)) + it * AT_TH) / (AT_DP / 8);
            const uint4 *p = reinterpret_cast<const uint4 *>(src + (long long)min(max(r, lo_), hi_ - 1) * src_stride + cc);
#pragma unroll
            for (int w = 0; w < W; ++w) raw[it][w] = p[w];
        }
    }
    __device__ __forceinline__ void commit(bf16_t *lds, int Dh) const {
        const int c = ((threadIdx.x & (AT_TH - 1)) % (AT_DP / 8)) * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = (int)((threadIdx.x & (AT_TH - 1)) + it * AT_TH) / (AT_DP / 8), r = first_row + rr;
            if (sizeof(T) == 2 && (Dh % 8) == 0) {   // bf16 source, whole 16-byte pieces: straight to LDS, no fp32 round trip
                const bool ok = r >= lo && r < hi && c < Dh;
                uint4 w = raw[it][0];
                if (!ok) w = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4 *>(lds + rr * AT_LD + c) = w;
                continue;
            }
            float v[8];
            if (sizeof(T) == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned wv = (&raw[it][0].x)[q];
                    v[2 * q] = __uint_as_float(wv << 16);
                    v[2 * q + 1] = __uint_as_float(wv & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __uint_as_float((&raw[it][0].x)[q]);
            }
            const bool row_ok = r >= lo && r < hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (row_ok && c + j < Dh) ? v[j] : 0.f;
            st8(lds + rr * AT_LD + c, v);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(AT_TH) void relpos_attn_fwd_kernel(const T *__restrict__ qkv, const T *__restrict__ pk,
                                                              const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                              const int32_t *__restrict__ key_lens, T *__restrict__ out,
                                                              float *__restrict__ lse, int Tn, int H, int Dh, float scale,
                                                              int causal, float pdrop, unsigned long long seed,
                                                              const unsigned long long *__restrict__ seed_dev,
                                                              int nparts, int part_keys, float *__restrict__ part_o, float *__restrict__ part_ml) {
    // nparts > 1 (long sequences, few utterances): blockIdx.x = query block * nparts + part; a part covers part_keys keys and leaves its
    // un-normalised (O, m, l) in part_o / part_ml for relpos_attn_merge_kernel - the query blocks of a causal T' = 4000 utterance need
    // 2 .. 63 key tiles each and there are only 128 of them per utterance: the longest set the time and half the CUs stood idle
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    bf16_t *k_lds = reinterpret_cast<bf16_t *>(smem);             // [AT_KT][AT_LD]
    bf16_t *v_lds = k_lds + AT_KT * AT_LD;                        // [AT_KT][AT_LD]
    bf16_t *p_lds = v_lds + AT_KT * AT_LD;                        // [AT_BAND][AT_LD]
    float *g_all = reinterpret_cast<float *>(p_lds + AT_BAND * AT_LD);  // [4 waves][64][32]
    const int part = nparts > 1 ? (int)(blockIdx.x % nparts) : 0;
    const int b = blockIdx.z, h = blockIdx.y, i0 = (nparts > 1 ? (int)(blockIdx.x / nparts) : (int)blockIdx.x) * AT_QB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    float *g_lds = g_all + wave * 64 * 32;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;  // qkv row (b,t) stride
    const T *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const int iq = i0 + wave * AT_QW + r;          // this lane's query row
    const int iqc = min(iq, Tn - 1);

    // Q + u, Q + v as B operands: lane (i = r, hh) holds dims 16s + 8hh + [0,8)
    bf16x8 qu[4], qv[4];
    {
        const bool fast = (Dh % 8) == 0;
        float q8[4][8], u8[4][8], v8[4][8];
        auto fetch = [&](auto fast_tag) {      // all 12 pieces in flight together (masked below)
            constexpr bool F = decltype(fast_tag)::value;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                load8_raw<T, F>(q_base + (long long)iqc * row_stride, 16 * s + 8 * hh, Dh, q8[s]);
                load8_raw<float, F>(bias_u + h * Dh, 16 * s + 8 * hh, Dh, u8[s]);
                load8_raw<float, F>(bias_v + h * Dh, 16 * s + 8 * hh, Dh, v8[s]);
            }
        };
        if (fast) fetch(std::true_type{});
        else fetch(std::false_type{});
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = 16 * s + 8 * hh + j < Dh;
                qu[s][j] = (bf16_t)(ok ? q8[s][j] + u8[s][j] : 0.f);
                qv[s][j] = (bf16_t)(ok ? q8[s][j] + v8[s][j] : 0.f);
            }
    }
    f32x16 o_acc[2];
    o_acc[0] = (f32x16){0};
    o_acc[1] = (f32x16){0};
    float m_run = -INFINITY, l_run = 0.f;
    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = drop_scale16(thr);
    const unsigned row_state = attn_row_state((unsigned long long)(b * H + h) * Tn + iq, drop_key(seed));

    int j_end = len;
    if (causal) j_end = min(j_end, causal_limit(i0 + AT_QB - 1, causal) + 1);  // keys beyond the last query's limit are never attended
    const int j_begin = part * part_keys;
    if (nparts > 1) {
        j_end = min(j_end, j_begin + part_keys);
        if (j_begin >= j_end) return;                                          // this part has no keys for this query block (workgroup-uniform)
    }
    const bool pipe = (Dh % 8) == 0;   // 16-byte aligned row pieces: tiles are requested one iteration ahead (StagePieces)
    StagePieces<T, AT_KT> sk, sv;
    StagePieces<T, AT_BAND> sp;
#ifdef AT_PROFILE
    long long facc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_prev = clock64();
#define ATF_STAMP(i) do { const long long n_ = clock64(); facc[i] += n_ - ft_prev; ft_prev = n_; } while (0)
#else
#define ATF_STAMP(i)
#endif
    if (pipe && j_end > 0) {
        sk.request(q_base + Dh, row_stride, j_begin, 0, Tn, Dh);
        sv.request(q_base + 2 * Dh, row_stride, j_begin, 0, Tn, Dh);
        sp.request(pk + (long long)h * Dh, D, j_begin - i0 - (AT_QB - 1) + Tn - 1, 0, 2 * Tn - 1, Dh);
    }
    ATF_STAMP(0);   // prologue
    for (int j0 = j_begin; j0 < j_end; j0 += AT_KT) {
        __syncthreads();  // previous tile fully consumed
        if (pipe) {
            sk.commit(k_lds, Dh);
            sv.commit(v_lds, Dh);
            sp.commit(p_lds, Dh);
            if (j0 + AT_KT < j_end) {   // next tile: in flight during this tile's MFMAs
                const int jn = j0 + AT_KT;
                sk.request(q_base + Dh, row_stride, jn, 0, Tn, Dh);
                sv.request(q_base + 2 * Dh, row_stride, jn, 0, Tn, Dh);
                sp.request(pk + (long long)h * Dh, D, jn - i0 - (AT_QB - 1) + Tn - 1, 0, 2 * Tn - 1, Dh);
            }
        } else {
            stage_rows<T, AT_KT>(k_lds, q_base + Dh, row_stride, j0, 0, Tn, Dh);
            stage_rows<T, AT_KT>(v_lds, q_base + 2 * Dh, row_stride, j0, 0, Tn, Dh);
            // band row R <-> r = j0 - i0 - (AT_QB - 1) + Tn - 1 + R
            stage_rows<T, AT_BAND>(p_lds, pk + (long long)h * Dh, D, j0 - i0 - (AT_QB - 1) + Tn - 1, 0, 2 * Tn - 1, Dh);
        }
        __syncthreads();
        ATF_STAMP(1);   // staging
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + 32 * sub;
            if (jb >= j_end) break;                                       // wave-uniform
            if (causal && jb > causal_limit(i0 + wave * AT_QW + 31, causal)) break;              // whole sub-block is in the future of this wave
            // ---- AC^T: rows = keys, col = query
            f32x16 s_acc = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(k_lds + (32 * sub + r) * AT_LD + 16 * s + 8 * hh);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qu[s], s_acc, 0, 0, 0);
            }
            // ---- G^T = Pband . (Q+v)^T : band rows base + [0,64)
            const int base = 32 * sub - 32 * wave + (AT_QB - AT_QW);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x16 g_acc = {0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(p_lds + (base + 32 * rb + r) * AT_LD + 16 * s + 8 * hh);
                    g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qv[s], g_acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) g_lds[(32 * rb + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = g_acc[g];
            }
            __builtin_amdgcn_wave_barrier();
            ATF_STAMP(2);   // AC + G MFMAs, G tile to LDS
            // ---- scores for this lane's query: 16 keys j = jb + (g&3) + 8(g>>2) + 4hh ; BD via the skewed read
            float sc[16], bdv[16];
            float mx = -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g) bdv[g] = g_lds[((g & 3) + 8 * (g >> 2) + 4 * hh - r + 31) * 32 + r];   // unconditional, back to back
            pin_all(bdv);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int jl = (g & 3) + 8 * (g >> 2) + 4 * hh;
                const float bd = bdv[g];
                const int j = jb + jl;
                float x = (s_acc[g] + bd) * scale;
                if (j >= len || (causal && j > causal_limit(iq, causal))) x = -INFINITY;
                sc[g] = x;
                mx = fmaxf(mx, x);
            }
            __builtin_amdgcn_wave_barrier();
            mx = fmaxf(mx, other_half(mx));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
            float psum = 0.f;
            bf16x8 pb[2];
            unsigned km[4] = {0xfu, 0xfu, 0xfu, 0xfu};   // keep-bits of this lane's four runs of four consecutive keys
            if (pdrop > 0.f) {
                const unsigned kw = attn_keep16(row_state, jb >> 5, hh, thr);      // the attention dropout stream (csrc/attn_common.h)
#pragma unroll
                for (int q = 0; q < 4; ++q) km[q] = (kw >> (4 * q)) & 0xfu;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float p = (m_new == -INFINITY) ? 0.f : __expf(sc[g] - m_new);
                psum += p;
                if (pdrop > 0.f) p = ((km[g >> 2] >> (g & 3)) & 1u) ? p * keep_scale : 0.f;
                pb[g >> 3][g & 7] = (bf16_t)p;
            }
            psum += other_half(psum);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; ++g) o_acc[db][g] *= alpha;
            ATF_STAMP(3);   // skewed read, softmax update, P fragments, O rescale
            // ---- O^T += V^T . P^T ; A = V^T through the transposing LDS read
            const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int kbase = 32 * sub + 16 * s + 4 * hh;
                    const bf16_t *a0 = v_lds + (kbase + q4) * AT_LD + 32 * db + 16 * mhalf + 4 * p4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0 + 8 * AT_LD));
                    bf16x8 va;
                    va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
                    va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[s], o_acc[db], 0, 0, 0);
                }
            }
        }
    }
    ATF_STAMP(4);       // (P.V of the last sub-block; earlier ones are folded into phase 2 of the next)
    if (nparts > 1) {   // un-normalised partial result of this key range: merged by relpos_attn_merge_kernel
        if (iq < Tn) {
            const size_t row = (((size_t)b * H + h) * Tn + iq) * nparts + part;
            float *po = part_o + row * AT_DP;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(po + 32 * db + 8 * q + 4 * hh) =
                        make_float4(o_acc[db][4 * q], o_acc[db][4 * q + 1], o_acc[db][4 * q + 2], o_acc[db][4 * q + 3]);
            if (hh == 0) *reinterpret_cast<float2 *>(part_ml + row * 2) = make_float2(m_run, l_run);
        }
        return;
    }
    // ---- epilogue: out[b, iq, h*Dh + d] = O / l ; lse = m + log l
    if (iq < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        T *orow = out + ((long long)b * Tn + iq) * D + (long long)h * Dh;
        // four consecutive head dims per store (the accumulator registers 4q .. 4q+3 of a lane are consecutive dims): 8 stores of
        // 8 bytes instead of 32 of 2 - the 2-byte version was 17 % of the kernel, store-issue bound
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * db + 8 * q + 4 * hh;
                if (d + 4 <= Dh && (Dh % 4) == 0)
                    st4(orow + d, o_acc[db][4 * q] * inv, o_acc[db][4 * q + 1] * inv, o_acc[db][4 * q + 2] * inv, o_acc[db][4 * q + 3] * inv);
                else
                    for (int e = 0; e < 4; ++e)
                        if (d + e < Dh) st1(orow + d + e, o_acc[db][4 * q + e] * inv);
            }
        if (hh == 0 && lse) lse[((long long)b * H + h) * Tn + iq] = m_run + __logf(l_run);
    }
#ifdef AT_PROFILE
    ATF_STAMP(5);       // epilogue
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 64 && lse)
        for (int i = 0; i < 6; ++i) reinterpret_cast<long long *>(lse)[i] = facc[i];   // (profile build: clobbers the first lse values)
#endif
}



// out[b, i, h, :] = sum_p e^(m_p - M) O_p / sum_p e^(m_p - M) l_p over the key parts that exist for query i (part p starts at key
// p * part_keys; a query reaches keys < min(len, causal limit + 1)), lse = M + log(sum). 16 threads per (b, h, i): four head dims each.
template <typename T>
__global__ __launch_bounds__(256) void relpos_attn_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                                const int32_t *__restrict__ key_lens, T *__restrict__ out,
                                                                float *__restrict__ lse, int B, int Tn, int H, int Dh, int causal,
                                                                int nparts, int part_keys) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid >> 4;
    const int d = (int)(gid & 15) * 4;
    if (row >= (long long)B * H * Tn) return;
    const int i = (int)(row % Tn), h = (int)((row / Tn) % H), b = (int)(row / ((long long)Tn * H));
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    // parts are launched per query BLOCK: a part exists for this row iff it has keys for the block's last query
    const int i_last = min((i / AT_QB) * AT_QB + AT_QB - 1, Tn - 1 + AT_QB);
    int j_end = len;
    if (causal) j_end = min(j_end, causal_limit(i_last, causal) + 1);
    const int np = min(nparts, (j_end + part_keys - 1) / part_keys);
    const float *ml = part_ml + row * nparts * 2;
    float M = -INFINITY;
    for (int p = 0; p < np; ++p) M = fmaxf(M, ml[2 * p]);
    float L = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < np; ++p) {
        const float m = ml[2 * p], w = (m == -INFINITY) ? 0.f : __expf(m - M);
        L += w * ml[2 * p + 1];
        const float4 v = *reinterpret_cast<const float4 *>(part_o + (row * nparts + p) * AT_DP + d);
        o[0] += w * v.x; o[1] += w * v.y; o[2] += w * v.z; o[3] += w * v.w;
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    T *orow = out + ((long long)b * Tn + i) * (H * Dh) + (long long)h * Dh;
    if (d + 4 <= Dh && (Dh % 4) == 0) st4(orow + d, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    else
        for (int e = 0; e < 4; ++e)
            if (d + e < Dh) st1(orow + d + e, o[e] * inv);
    if (d == 0 && lse) lse[row] = M + __logf(L);
}

// =====================================================================================================================
// Forward for short sequences (T <= 256: the mixture encoder's T' = 250 and the speaker encoder's 125 at BASELINE configs[1]),
// bf16, Dh = 64. The streaming kernel above runs ONE wave per SIMD there (256 workgroups x 4 waves on 256 CUs) through four
// key tiles of [stage through registers -> barrier -> MFMA -> G tile through LDS -> softmax]: stamped, 49 % of a wave's cycles
// sit in the softmax phase, 24 % in staging, 22 % around the MFMAs - serial latencies that nothing else on the SIMD covers. Here
//   * everything a workgroup needs is staged ONCE by LDS-DMA (inline-asm global_load_lds_dwordx4, see csrc/wgrad.hip): all K and V
//     rows of the (b, h) pair, and the band of positional rows its queries can reach - 32 + 32 + 48 KiB at T' = 250;
//   * a workgroup is 8 waves = (query blocks of 32) x (key parts): 4 x 2 for 128 queries (T' > 128), 2 x 4 for 64 queries; two
//     waves per SIMD cover each other's LDS / transcendental latencies; the key parts of a query block are merged at the end
//     through LDS with the usual (m, l, O) rescale;
//   * the band products are ROLLED: the 64 band rows a 32 x 32 block of scores needs overlap the next block's by 32, so each
//     sub-block computes one new 32-row G block (4 MFMAs) instead of two; G tiles live in LDS as fp16 (4 KiB per wave: the
//     rounding, 2^-11 relative, is far below that of the bf16 operands).
// Same arithmetic, masks, dropout stream and outputs as relpos_attn_fwd_kernel.
// =====================================================================================================================

template <int QH>   // queries per workgroup: 128 (keys padded to 256, 2 key parts) or 64 (keys padded to 128, 4 key parts)
__global__ __launch_bounds__(512, 2) void relpos_attn_fwd_short_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ pk,
                                                                       const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                       const int32_t *__restrict__ key_lens, bf16_t *__restrict__ out,
                                                                       float *__restrict__ lse, int Tn, int H, float scale, int causal,
                                                                       float pdrop, unsigned long long seed,
                                                                       const unsigned long long *__restrict__ seed_dev,
                                                                       unsigned short *__restrict__ keepbits /*[B*H*T][2][8] or NULL*/) {
    constexpr int Dh = 64, NQB = QH / 32, NKP = 8 / NQB, TPAD = 2 * QH, KP = TPAD / NKP, NB = QH + TPAD;   // NB band rows staged
    constexpr int K_OFF = 0, V_OFF = TPAD * 128, P_OFF = 2 * TPAD * 128, G_OFF = P_OFF + NB * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * QH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int qb = wave % NQB, kp = wave / NQB;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const bf16_t *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

#ifdef AT_PROFILE
    long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#define ATS_STAMP(i) do { const long long n_ = clock64(); sacc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define ATS_STAMP(i)
#endif
    // ---- stage K, V (all keys) and the band: piece = 8 rows x 128 B; LDS slot (row, pos) <- global chunk pos ^ swizzle(row)
    {
        const int prow = lane >> 3, pos = lane & 7;
        const int r_lo = (Tn - 1) - (i0 + QH - 1);
        for (int pc = wave; pc < (2 * TPAD + NB) / 8; pc += 8) {
            const void *g;
            unsigned dst;
            if (pc < TPAD / 8) {                     // K: row-wise b128 fragment reads
                const int row = pc * 8 + prow;
                g = q_base + (long long)min(row, Tn - 1) * row_stride + Dh + ((pos ^ ((row >> 1) & 7)) << 3);
                dst = K_OFF + pc * 1024;
            } else if (pc < 2 * TPAD / 8) {          // V: transposing reads (k = key)
                const int pv = pc - TPAD / 8, row = pv * 8 + prow;
                g = q_base + (long long)min(row, Tn - 1) * row_stride + 2 * Dh + ((pos ^ (((row >> 1) & 1) << 2)) << 3);
                dst = V_OFF + pv * 1024;
            } else {                                 // band rows r_lo + R, clamped into the table (out-of-table rows only meet masked keys)
                const int pp = pc - 2 * TPAD / 8, row = pp * 8 + prow;
                g = pk + (long long)min(max(r_lo + row, 0), 2 * Tn - 2) * D + (long long)h * Dh + ((pos ^ ((row >> 1) & 7)) << 3);
                dst = P_OFF + pp * 1024;
            }
            at_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + dst));
        }
    }
    // ---- this lane's query: Q + u, Q + v as B operands (dims 16s + 8hh + [0,8))
    const int iq = i0 + 32 * qb + r, iqc = min(iq, Tn - 1);
    bf16x8 qu[4], qv[4];
    {
        float q8[4][8], u8[4][8], v8[4][8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ld8(q_base + (long long)iqc * row_stride + 16 * s + 8 * hh, q8[s]);
            ld8(bias_u + h * Dh + 16 * s + 8 * hh, u8[s]);
            ld8(bias_v + h * Dh + 16 * s + 8 * hh, v8[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                qu[s][j] = (bf16_t)(q8[s][j] + u8[s][j]);
                qv[s][j] = (bf16_t)(q8[s][j] + v8[s][j]);
            }
    }
    ATS_STAMP(0);   // DMA issue + q loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ATS_STAMP(1);   // wait for the staged tiles

    f32x16 o_acc[2];
    o_acc[0] = (f32x16){0};
    o_acc[1] = (f32x16){0};
    float m_run = -INFINITY, l_run = 0.f;
    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = drop_scale16(thr);
    const unsigned row_state = attn_row_state((unsigned long long)(b * H + h) * Tn + iq, drop_key(seed));
    const int lim_q = causal ? causal_limit(iq, causal) : 0x3fffffff;
    const int lim_blk = causal ? causal_limit(min(i0 + 32 * qb + 31, Tn - 1), causal) : 0x3fffffff;   // last key any query of this wave attends
    const int j_lim = min(len - 1, lim_q);      // last key this lane's query attends ...
    int j_all = j_lim;                          // ... and the last one EVERY query of the wave attends
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) j_all = min(j_all, __shfl_xor(j_all, o));
    j_all = __builtin_amdgcn_readfirstlane(j_all);
    const unsigned ks_bits = __float_as_uint(keep_scale);
    const char *k_lds = smem + K_OFF, *v_lds = smem + V_OFF, *p_lds = smem + P_OFF;
    _Float16 *g_lds = reinterpret_cast<_Float16 *>(smem + G_OFF + wave * 4096);   // two slots of [32 band rows][32 queries]
    const int fr_swz = (r >> 1) & 7;   // fragment rows are 32-aligned + r: the swizzle term of a b128 fragment read is the lane's own
    const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;

    auto g_block = [&](int Rblk, int slot) {   // G^T rows Rblk .. Rblk+31 = Pband . (Q+v)^T  -> fp16 slot
        f32x16 g_acc = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(p_lds + (Rblk + r) * 128 + (((2 * s + hh) ^ fr_swz) << 4));
            g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qv[s], g_acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) g_lds[slot * 1024 + ((g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = (_Float16)g_acc[g];
    };
    const int j_first = kp * KP;
    int j_last = min(min((kp + 1) * KP, len), lim_blk + 1);    // keys [j_first, j_last) are live for this wave
    if (i0 + 32 * qb >= Tn) j_last = j_first;                   // a query block beyond the sequence
    const int nsub = j_last > j_first ? (j_last - j_first + 31) / 32 : 0;
    if (nsub > 0) g_block(j_first - 32 * qb + QH - 32, 0);
    for (int sub = 0; sub < nsub; ++sub) {
        const int jb = j_first + 32 * sub;
        const int Rb = jb - 32 * qb + QH - 32;      // band block A = rows Rb.. (slot sub & 1), block B = rows Rb + 32.. (slot (sub + 1) & 1)
        f32x16 s_acc = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(k_lds + (jb + r) * 128 + (((2 * s + hh) ^ fr_swz) << 4));
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qu[s], s_acc, 0, 0, 0);
        }
        g_block(Rb + 32, (sub + 1) & 1);
        __builtin_amdgcn_wave_barrier();
        ATS_STAMP(2);   // AC + G MFMAs, G store
        float sc[16], bdv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {   // all 16 skewed reads issued back to back, unconditionally
            const int jl = (g & 3) + 8 * (g >> 2) + 4 * hh, rowq = jl - r + 31;
            bdv[g] = (float)g_lds[(((rowq >> 5) ^ sub) & 1) * 1024 + (rowq & 31) * 32 + r];
        }
        pin_all(bdv);
        // full: every key of this sub-block exists for every query of the wave (all but the last sub-block of an unmasked sequence):
        // no key mask, and the running maximum is finite
        const bool full = jb + 31 <= j_all;
        auto scores = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = jb + (g & 3) + 8 * (g >> 2) + 4 * hh;
                float x = (s_acc[g] + bdv[g]) * scale;
                if (!FULL && j > j_lim) x = -INFINITY;
                sc[g] = x;
                mx = fmaxf(mx, x);
            }
        };
        if (full) scores(std::true_type{});
        else scores(std::false_type{});
        __builtin_amdgcn_wave_barrier();
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
        float psum = 0.f;
        bf16x8 pb[2];
        unsigned kw = 0xffffu;   // keep-bits of this lane's 16 keys (bit g = accumulator element g)
        if (pdrop > 0.f) {
            kw = attn_keep16(row_state, jb >> 5, hh, thr);
            // kept for the backward (tsasr_relpos_attn_keepbits): it then reads one 16-byte word per query row instead of hashing again
            // (the hashes are ~5 us of its 52 us at T' = 250)
            if (keepbits && iq < Tn) keepbits[(((size_t)(b * H + h) * Tn + iq) * 2 + hh) * 8 + (jb >> 5)] = (unsigned short)kw;
        }
        float pf[16];
        auto probs = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float p = __expf(sc[g] - m_new);
                if (!FULL) p = (m_new == -INFINITY) ? 0.f : p;
                psum += p;
                // dropout: the element's keep-bit spread over a word (v_bfe_i32) ANDed with the bits of 1 / (1 - p)
                if (pdrop > 0.f) p *= __uint_as_float((unsigned)((int)(kw << (31 - g)) >> 31) & ks_bits);
                pf[g] = p;
            }
        };
        if (full) probs(std::true_type{});
        else probs(std::false_type{});
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            pb[s2] = bf16x8_of(pk_bf16(pf[8 * s2], pf[8 * s2 + 1]), pk_bf16(pf[8 * s2 + 2], pf[8 * s2 + 3]), pk_bf16(pf[8 * s2 + 4], pf[8 * s2 + 5]),
                               pk_bf16(pf[8 * s2 + 6], pf[8 * s2 + 7]));
        psum += other_half(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) o_acc[db][g] *= alpha;
        ATS_STAMP(3);   // skewed read, softmax, P fragments, O rescale
        // O^T += V^T . P^T ; A = V^T through the transposing read (k order of pb = accumulator row order)
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
    }
    ATS_STAMP(4);   // P.V
    // ---- merge the key parts of each query block: parts kp > 0 hand (m, l, O) to part 0 through LDS (K / V / band are dead now)
    __syncthreads();
    ATS_STAMP(5);   // barrier before the merge
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
        const float a_me = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new), a_o = (m_o == -INFINITY) ? 0.f : __expf(m_o - m_new);
        l_run = l_run * a_me + l_o * a_o;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                o_acc[db][g] = o_acc[db][g] * a_me + oth[(32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] * a_o;
    }
    if (iq < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        bf16_t *orow = out + ((long long)b * Tn + iq) * D + (long long)h * Dh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * db + 8 * q + 4 * hh;
                st4(orow + d, o_acc[db][4 * q] * inv, o_acc[db][4 * q + 1] * inv, o_acc[db][4 * q + 2] * inv, o_acc[db][4 * q + 3] * inv);
            }
        if (hh == 0 && lse) lse[((long long)b * H + h) * Tn + iq] = m_run + __logf(l_run);
    }
#ifdef AT_PROFILE
    ATS_STAMP(6);   // merge + epilogue
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0 && lse)
        for (int i = 0; i < 7; ++i) reinterpret_cast<long long *>(lse)[i] = sacc[i];   // (profile build: clobbers the first lse values)
#endif
}

// =====================================================================================================================
// Backward. Two deterministic passes that both recompute the probabilities from (q,k,p,lse) - no T x T tensor, no atomics:
//   bwd_q  (query-major, same orientation as the forward): dQ = dQ_ac + dQ_bd, per-workgroup partial sums of the gradients
//          of pos_bias_u / pos_bias_v, and the shifted-back score gradient dBD[h][r][b][i] = scale*dS[i, j=r+i-T+1] (compute
//          dtype) from which the host obtains d(pk) with one library GEMM per head (d(pk) is a parameter-like reduction
//          over the batch).
//   bwd_kv (key-major: a lane owns one KEY, accumulators hold dK^T / dV^T): dK, dV.
// dS = P * (dP - delta), delta_i = dO_i . O_i ; with dropout P_d = P*m/(1-p): dP = (dO.V^T)*m/(1-p), same counter-based mask.
// =====================================================================================================================
// KG = 2: 8 waves - two groups of four share the workgroup's 128 queries and walk one half of its key tiles each (own K / V / band
// tiles and G scratch in LDS), the second group hands its dQ shares to the first through LDS at the end. At B*H*T'/128 <= CUs there is
// one workgroup per CU and with KG = 1 one wave per SIMD: nothing covers the softmax VALU work, the LDS round trips or the staging.
template <typename T, int KG>
__global__ __launch_bounds__(AT_TH * KG) void relpos_attn_bwd_q_kernel(const T *__restrict__ qkv, const T *__restrict__ pk,
                                                                const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                                const int32_t *__restrict__ key_lens, const T *__restrict__ out,
                                                                const T *__restrict__ dout, const float *__restrict__ lse,
                                                                T *__restrict__ dqkv, T *__restrict__ pd_out, T *__restrict__ ds_out,
                                                                float *__restrict__ slab_uv, int Tp,
                                                                int Bn, int Tn, int H, int Dh, float scale, int causal, float pdrop,
                                                                unsigned long long seed, const unsigned long long *__restrict__ seed_dev,
                                                                int nparts, int part_keys, float *__restrict__ dq_part,
                                                                const unsigned short *__restrict__ keepbits /*forward's keep-bits or NULL*/) {
    // nparts > 1: blockIdx.x = query block * nparts + key part (see relpos_attn_fwd_kernel); a part leaves its share of dQ in fp32 in
    // dq_part (summed by relpos_attn_dq_merge_kernel) and its own row of pos_bias partial sums; P_d / dS columns are disjoint anyway
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (seed_dev) seed += *seed_dev;
    constexpr size_t GROUP_BYTES = (size_t)(2 * AT_KT + AT_BAND) * AT_LD * sizeof(bf16_t) + (size_t)AT_NW * 64 * 32 * sizeof(float);
    const int kgrp = KG > 1 ? (int)(threadIdx.x >> 8) : 0;       // wave-uniform
    bf16_t *k_lds = reinterpret_cast<bf16_t *>(smem + kgrp * GROUP_BYTES);
    bf16_t *v_lds = k_lds + AT_KT * AT_LD;
    bf16_t *p_lds = v_lds + AT_KT * AT_LD;
    float *g_all = reinterpret_cast<float *>(p_lds + AT_BAND * AT_LD);
    const int part = nparts > 1 ? (int)(blockIdx.x % nparts) : 0;
    const int b = blockIdx.z, h = blockIdx.y, i0 = (nparts > 1 ? (int)(blockIdx.x / nparts) : (int)blockIdx.x) * AT_QB;
    const int tid = threadIdx.x & (AT_TH - 1), lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    float *g_lds = g_all + wave * 64 * 32;
    const int D = H * Dh, R = 2 * Tn - 1;
    const long long row_stride = 3LL * D;
    const T *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const int iq = i0 + wave * AT_QW + r, iqc = min(iq, Tn - 1);
    const bool q_ok = iq < Tn;

    bf16x8 qu[4], qv[4], dob[4];
    float delta = 0.f;
    const T *orow = out + ((long long)b * Tn + iqc) * D + (long long)h * Dh;
    const T *dorow = dout + ((long long)b * Tn + iqc) * D + (long long)h * Dh;
    {
        const bool fast = (Dh % 8) == 0;
        float q8[4][8], u8[4][8], v8[4][8], d8[4][8], o8[4][8];
        auto fetch = [&](auto fast_tag) {      // all 20 pieces in flight together (masked below)
            constexpr bool F = decltype(fast_tag)::value;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                load8_raw<T, F>(q_base + (long long)iqc * row_stride, 16 * s + 8 * hh, Dh, q8[s]);
                load8_raw<float, F>(bias_u + h * Dh, 16 * s + 8 * hh, Dh, u8[s]);
                load8_raw<float, F>(bias_v + h * Dh, 16 * s + 8 * hh, Dh, v8[s]);
                load8_raw<T, F>(dorow, 16 * s + 8 * hh, Dh, d8[s]);
                load8_raw<T, F>(orow, 16 * s + 8 * hh, Dh, o8[s]);
            }
        };
        if (fast) fetch(std::true_type{});
        else fetch(std::false_type{});
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = 16 * s + 8 * hh + j < Dh;
                const float dd = (q_ok && ok) ? d8[s][j] : 0.f, oo = (q_ok && ok) ? o8[s][j] : 0.f;
                qu[s][j] = (bf16_t)(ok ? q8[s][j] + u8[s][j] : 0.f);
                qv[s][j] = (bf16_t)(ok ? q8[s][j] + v8[s][j] : 0.f);
                dob[s][j] = (bf16_t)dd;
                delta += dd * oo;
            }
    }
    delta += other_half(delta);
    const float lse_i = lse[((long long)b * H + h) * Tn + iqc];
    // forward's keep-bits of this lane's query row: 8 sub-blocks x 16 bits, one unconditional 16-byte load (a stand-in address without them)
    const bool has_kb = keepbits != nullptr;
    const uint4 kbw = *reinterpret_cast<const uint4 *>(has_kb ? reinterpret_cast<const char *>(keepbits + (((size_t)(b * H + h) * Tn + iqc) * 2 + hh) * 8)
                                                             : reinterpret_cast<const char *>(qkv));
    f32x16 dqu[2], dqv[2];
    dqu[0] = dqu[1] = dqv[0] = dqv[1] = (f32x16){0};
    const unsigned thr = drop_thr16(pdrop);
    const float keep_scale = drop_scale16(thr);
    const unsigned row_state = attn_row_state((unsigned long long)(b * H + h) * Tn + iq, drop_key(seed));
    const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;

    // dG^T fragment masks: element (band row rl = 16 sp + 8 hh + e, query r) of a 32-key sub-block exists iff 0 <= rl + r - 31 < 32; as bf16-pair
    // bit masks (one AND per converted pair instead of a compare-select per element)
    unsigned dgmask[4][4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl0 = 16 * sp + 8 * hh + 2 * k + r - 31;
            dgmask[sp][k] = ((jl0 >= 0 && jl0 < 32) ? 0xffffu : 0u) | ((jl0 + 1 >= 0 && jl0 + 1 < 32) ? 0xffff0000u : 0u);
        }
    // this lane's row of P_d / dS, at its first key of a sub-block
    T *const prow0 = pd_out + (((long long)b * H + h) * Tn + iqc) * Tp + 4 * hh;
    T *const srow0 = ds_out + (((long long)b * H + h) * Tn + iqc) * Tp + 4 * hh;
    const unsigned ks_bits = __float_as_uint(keep_scale);
    // last key this lane's query attends (-1: a row beyond the sequence - its P_d, dS and everything derived from them are zero)
    const int j_max = q_ok ? min(len - 1, causal ? causal_limit(iq, causal) : 0x3fffffff) : -1;
    int j_all = j_max;      // ... and the last key EVERY query of this wave attends: sub-blocks up to it need no mask at all
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) j_all = min(j_all, __shfl_xor(j_all, o));
    j_all = __builtin_amdgcn_readfirstlane(j_all);
    int j_end = len;
    if (causal) j_end = min(j_end, causal_limit(i0 + AT_QB - 1, causal) + 1);
    int j_begin = part * part_keys;
    if (nparts > 1) j_end = min(j_end, j_begin + part_keys);   // an empty part (j_begin >= j_end) skips the loop and writes zero sums below
    // trips of the key loop: the same for every wave of the workgroup (the loop holds workgroup barriers); KG = 2: each group takes
    // `trips` tiles of the range, a group that runs out of keys idles through the remaining barriers
    const int ntile = max(0, (j_end - j_begin + AT_KT - 1) / AT_KT), trips = (ntile + KG - 1) / KG;
    if (KG > 1) {
        j_begin += kgrp * trips * AT_KT;
        j_end = min(j_end, j_begin + trips * AT_KT);
    }
    const bool pipe = KG == 1 && (Dh % 8) == 0;   // KG = 2: the other group's waves cover the round trips; the prefetch registers would spill
    StagePieces<T, AT_KT> sk, sv;
    StagePieces<T, AT_BAND> sp;
#ifdef AT_PROFILE
    long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define AT_STAMP(i) do { const long long n_ = clock64(); acc_t[i] += n_ - t_prev; t_prev = n_; } while (0)
#else
#define AT_STAMP(i)
#endif
    if (pipe && j_end > j_begin) {
        sk.request(q_base + Dh, row_stride, j_begin, 0, Tn, Dh);
        sv.request(q_base + 2 * Dh, row_stride, j_begin, 0, Tn, Dh);
        sp.request(pk + (long long)h * Dh, D, j_begin - i0 - (AT_QB - 1) + Tn - 1, 0, R, Dh);
    }
    AT_STAMP(0);   // prologue (q, do, o loads; first tile requests)
    for (int trip = 0; trip < trips; ++trip) {
        const int j0 = j_begin + trip * AT_KT;
        const bool idle = j0 >= j_end;             // group-uniform (KG = 2 only): no keys left for this group
        __syncthreads();
        if (idle) { __syncthreads(); continue; }
        const int r_first = j0 - i0 - (AT_QB - 1) + Tn - 1;
        if (pipe) {
            sk.commit(k_lds, Dh);
            sv.commit(v_lds, Dh);
            sp.commit(p_lds, Dh);
            if (j0 + AT_KT < j_end) {
                const int jn = j0 + AT_KT;
                sk.request(q_base + Dh, row_stride, jn, 0, Tn, Dh);
                sv.request(q_base + 2 * Dh, row_stride, jn, 0, Tn, Dh);
                sp.request(pk + (long long)h * Dh, D, jn - i0 - (AT_QB - 1) + Tn - 1, 0, R, Dh);
            }
        } else {
            stage_rows<T, AT_KT>(k_lds, q_base + Dh, row_stride, j0, 0, Tn, Dh);
            stage_rows<T, AT_KT>(v_lds, q_base + 2 * Dh, row_stride, j0, 0, Tn, Dh);
            stage_rows<T, AT_BAND>(p_lds, pk + (long long)h * Dh, D, r_first, 0, R, Dh);
        }
        __syncthreads();
        AT_STAMP(1);   // staging: wait for the tile, LDS stores, next requests, two barriers
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + 32 * sub;
            if (jb >= j_end) break;
            if (causal && jb > causal_limit(i0 + wave * AT_QW + 31, causal)) break;
            f32x16 s_acc = {0}, dpd = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(k_lds + (32 * sub + r) * AT_LD + 16 * s + 8 * hh);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qu[s], s_acc, 0, 0, 0);
                const bf16x8 va = *reinterpret_cast<const bf16x8 *>(v_lds + (32 * sub + r) * AT_LD + 16 * s + 8 * hh);
                dpd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dob[s], dpd, 0, 0, 0);
            }
            const int base = 32 * sub - 32 * wave + (AT_QB - AT_QW);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x16 g_acc = {0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 pa = *reinterpret_cast<const bf16x8 *>(p_lds + (base + 32 * rb + r) * AT_LD + 16 * s + 8 * hh);
                    g_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, qv[s], g_acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) g_lds[(32 * rb + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = g_acc[g];
            }
            __builtin_amdgcn_wave_barrier();
            AT_STAMP(2);   // S, dP, G MFMAs + G tile to LDS
            float ds[16], pdv[16], bdv[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) bdv[g] = g_lds[((g & 3) + 8 * (g >> 2) + 4 * hh - r + 31) * 32 + r];   // unconditional, back to back
            pin_all(bdv);
            unsigned kw = 0xffffu;   // keep-bits of this lane's 16 keys (bit g = accumulator element g)
            if (pdrop > 0.f) {
                if (has_kb) {       // workgroup-uniform; jb >> 5 is wave-uniform
                    const int sb = jb >> 5;
                    const unsigned pair = (sb >> 1) == 0 ? kbw.x : (sb >> 1) == 1 ? kbw.y : (sb >> 1) == 2 ? kbw.z : kbw.w;
                    kw = (sb & 1) ? (pair >> 16) : (pair & 0xffffu);
                } else {
                    kw = attn_keep16(row_state, jb >> 5, hh, thr);
                }
            }
            auto probs = [&](auto full_tag) {       // full: every key of the sub-block exists for every query of the wave
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = jb + (g & 3) + 8 * (g >> 2) + 4 * hh;
                    float p = __expf((s_acc[g] + bdv[g]) * scale - lse_i);
                    if (!FULL) p = j > j_max ? 0.f : p;     // one compare: key length, look-ahead mask and "no such query"
                    // dropout factor: the element's keep-bit spread over a word (v_bfe_i32) ANDed with the bits of 1 / (1 - p)
                    const float keep = pdrop > 0.f ? __uint_as_float((unsigned)((int)(kw << (31 - g)) >> 31) & ks_bits) : 1.f;
                    pdv[g] = p * keep;
                    ds[g] = p * (dpd[g] * keep - delta) * scale;
                }
            };
            if (jb + 31 <= j_all) probs(std::true_type{});
            else probs(std::false_type{});
            // P_d and scale*dS of this lane's query, 4 consecutive keys per store: the key-major pass (dK, dV) and d(pk) read them
            if (q_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    st4(prow0 + jb + 8 * q, pdv[4 * q], pdv[4 * q + 1], pdv[4 * q + 2], pdv[4 * q + 3]);
                    st4(srow0 + jb + 8 * q, ds[4 * q], ds[4 * q + 1], ds[4 * q + 2], ds[4 * q + 3]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // inverse skew: dG^T[r_local = jl - i + 31][i] = dSs[i, jl]
            bf16x8 dsb[2];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int jl = (g & 3) + 8 * (g >> 2) + 4 * hh;
                g_lds[(jl - r + 31) * 32 + r] = ds[g];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                dsb[s2] = bf16x8_of(pk_bf16(ds[8 * s2], ds[8 * s2 + 1]), pk_bf16(ds[8 * s2 + 2], ds[8 * s2 + 3]), pk_bf16(ds[8 * s2 + 4], ds[8 * s2 + 5]),
                                    pk_bf16(ds[8 * s2 + 6], ds[8 * s2 + 7]));
            __builtin_amdgcn_wave_barrier();
            AT_STAMP(3);   // skewed read, p, dS, inverse skew
            // dQ_ac^T += K^T . dSs^T   (A = K^T through the transposing read; k order of dsb = accumulator row order)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16_t *a0 = k_lds + (32 * sub + 16 * s + 4 * hh + q4) * AT_LD + 32 * db + 16 * mhalf + 4 * p4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0 + 8 * AT_LD));
                    bf16x8 ka;
                    ka[0] = lo[0]; ka[1] = lo[1]; ka[2] = lo[2]; ka[3] = lo[3];
                    ka[4] = hi[0]; ka[5] = hi[1]; ka[6] = hi[2]; ka[7] = hi[3];
                    dqu[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, dsb[s], dqu[db], 0, 0, 0);
                }
            // dQ_bd^T += Pband^T . dG^T   (natural k order: band rows 16s' + 8hh + e), predicate = "this (r,i) has a key"
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                float dgv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dgv[e] = g_lds[(16 * sp + 8 * hh + e) * 32 + r];   // unconditional reads; elements without a key are masked after the conversion
                const bf16x8 dgb = bf16x8_of(pk_bf16(dgv[0], dgv[1]) & dgmask[sp][0], pk_bf16(dgv[2], dgv[3]) & dgmask[sp][1],
                                             pk_bf16(dgv[4], dgv[5]) & dgmask[sp][2], pk_bf16(dgv[6], dgv[7]) & dgmask[sp][3]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16_t *a0 = p_lds + (base + 16 * sp + 8 * hh + q4) * AT_LD + 32 * db + 16 * mhalf + 4 * p4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(a0 + 4 * AT_LD));
                    bf16x8 pa;
                    pa[0] = lo[0]; pa[1] = lo[1]; pa[2] = lo[2]; pa[3] = lo[3];
                    pa[4] = hi[0]; pa[5] = hi[1]; pa[6] = hi[2]; pa[7] = hi[3];
                    dqv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, dgb, dqv[db], 0, 0, 0);
                }
            }
            AT_STAMP(4);   // dQ MFMAs (transposing reads, dG fragments)
            __builtin_amdgcn_wave_barrier();
            AT_STAMP(5);
        }
    }
    if (KG > 1) {   // the second group's shares of dQ_ac / dQ_bd reach the first through LDS ([wave][register][lane]: conflict-free)
        __syncthreads();                                            // every tile consumed: the second group's area is free
        float *xch = reinterpret_cast<float *>(smem + GROUP_BYTES) + wave * 64 * 64;
        if (kgrp == 1) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    xch[(db * 16 + g) * 64 + lane] = dqu[db][g];
                    xch[(32 + db * 16 + g) * 64 + lane] = dqv[db][g];
                }
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                dqu[db][g] += xch[(db * 16 + g) * 64 + lane];
                dqv[db][g] += xch[(32 + db * 16 + g) * 64 + lane];
            }
    }
    // ---- dQ = dQ_ac + dQ_bd, four consecutive head dims per store (the three cases are workgroup-uniform: no per-store branches)
    T *dq = dqkv + ((long long)b * Tn + iqc) * row_stride + (long long)h * 3 * Dh;
    if (nparts > 1) {
        if (q_ok && ntile > 0) {
            float *dqp = dq_part + ((((size_t)b * H + h) * Tn + iq) * nparts + part) * AT_DP;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; g += 4)
                    *reinterpret_cast<float4 *>(dqp + 32 * db + 8 * (g >> 2) + 4 * hh) =
                        make_float4(dqu[db][g] + dqv[db][g], dqu[db][g + 1] + dqv[db][g + 1], dqu[db][g + 2] + dqv[db][g + 2], dqu[db][g + 3] + dqv[db][g + 3]);
        }
    } else if (Dh == AT_DP) {
        if (q_ok) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 16; g += 4)
                    st4(dq + 32 * db + 8 * (g >> 2) + 4 * hh, dqu[db][g] + dqv[db][g], dqu[db][g + 1] + dqv[db][g + 1], dqu[db][g + 2] + dqv[db][g + 2],
                        dqu[db][g + 3] + dqv[db][g + 3]);
        }
    } else if (q_ok) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int d = 32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh;
                if (d < Dh) st1(dq + d, dqu[db][g] + dqv[db][g]);
            }
    }
    // ---- partial sums over this wave's 32 queries for d(pos_bias_u) (= sum dQ_ac) and d(pos_bias_v) (= sum dQ_bd): the accumulators go
    // through the wave's scratch as [dim][query] and lane d adds up row d, starting at column d so that the lanes of a read fall into
    // different banks (64 DPP reductions with a guarded store each were 2,800 instructions = 15 % of this kernel's cycles). Rows beyond the
    // sequence hold exact zeros (j_max = -1 above): nothing to mask.
    float *slab = slab_uv + (((long long)(b * gridDim.x + blockIdx.x) * AT_NW + wave) * H + h) * 128;   // part = ((b, qtile), wave); row = [h][u 64 | v 64]
    float colsum[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) g_lds[(32 * db + (g & 3) + 8 * (g >> 2) + 4 * hh) * 32 + r] = which ? dqv[db][g] : dqu[db][g];
        __builtin_amdgcn_wave_barrier();
        float part_sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 32; ++k) part_sum[k & 3] += g_lds[lane * 32 + ((k + lane) & 31)];
        colsum[which] = (part_sum[0] + part_sum[1]) + (part_sum[2] + part_sum[3]);
    }
    slab[lane] = colsum[0];
    slab[64 + lane] = colsum[1];
#ifdef AT_PROFILE
    AT_STAMP(6);   // epilogue
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 64)
        for (int i = 0; i < 7; ++i) reinterpret_cast<long long *>(slab)[i] = acc_t[i];     // over this wave's own partial sums (row 1 * H of the slab)
#endif
}

// dQ[b, i, h, :] = sum over the key parts that exist for query i's block (fixed order) of the fp32 shares bwd_q left; 16 threads per row
template <typename T>
__global__ __launch_bounds__(256) void relpos_attn_dq_merge_kernel(const float *__restrict__ dq_part, const int32_t *__restrict__ key_lens,
                                                                   T *__restrict__ dqkv, int B, int Tn, int H, int Dh, int causal, int nparts,
                                                                   int part_keys) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid >> 4;
    const int d = (int)(gid & 15) * 4;
    if (row >= (long long)B * H * Tn) return;
    const int i = (int)(row % Tn), h = (int)((row / Tn) % H), b = (int)(row / ((long long)Tn * H));
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    int j_end = len;
    if (causal) j_end = min(j_end, causal_limit((i / AT_QB) * AT_QB + AT_QB - 1, causal) + 1);
    const int np = min(nparts, (j_end + part_keys - 1) / part_keys);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < np; ++p) {
        const float4 v = *reinterpret_cast<const float4 *>(dq_part + (row * nparts + p) * AT_DP + d);
        o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
    }
    T *dq = dqkv + ((long long)b * Tn + i) * (3LL * H * Dh) + (long long)h * 3 * Dh;
    if (d + 4 <= Dh && (Dh % 4) == 0) st4(dq + d, o[0], o[1], o[2], o[3]);
    else
        for (int e = 0; e < 4; ++e)
            if (d + e < Dh) st1(dq + d + e, o[e]);
}

// Key-major pass on the MATERIALISED probabilities: bwd_q leaves P_d (dropout applied) and scale*dS as [B,H,T,Tp] tensors (16 MB each
// at B=32, T'=250, bf16 - they stay in the 256 MB Infinity Cache between the two kernels), so dK and dV are two plain
// contractions over the queries,   dV^T[d][j] = sum_i dO[i][d] * P_d[i][j] ,   dK^T[d][j] = sum_i (q_i + u)[d] * dS[i][j] ,
// instead of a second recomputation of scores, band products and exponentials (the first version of this pass: 55 us per layer
// against 80 us for the query-major pass that had already computed every one of these numbers).
// workgroup = (b, h, 64 keys); wave = (32 keys) x (32 head dims); query chunks of 64 stream through LDS, all four operand
// fragments come out of it through the transposing read (k = query index is the row index of every tile).
#define KV2_LD 72
template <typename T>
__global__ __launch_bounds__(256) void relpos_attn_bwd_kv2_kernel(const T *__restrict__ qkv, const float *__restrict__ bias_u,
                                                                  const float *__restrict__ bias_v, const int32_t *__restrict__ key_lens,
                                                                  const T *__restrict__ dout, const T *__restrict__ pd,
                                                                  const T *__restrict__ ds, T *__restrict__ dqkv,
                                                                  T *__restrict__ qv_out /*[H][B*T][Dh] or NULL*/, int Tn, int Tp, int H,
                                                                  int Dh, int causal, int isplit, float *__restrict__ kv_part) {
    // isplit > 1 (long sequences in small batches): blockIdx.x = key tile * isplit + part; part p takes the query chunks i_begin + 64 (p + isplit n)
    // and leaves its fp32 share of (dK | dV) in kv_part [part][b][h][key][128] for relpos_attn_kv_merge_kernel. One workgroup per key tile walked
    // every query from its keys on - 63 chunks for the first tile of a causal T' = 4000 utterance, 1 for the last, the launch as long as the longest.
    __shared__ __attribute__((aligned(16))) bf16_t xp[64 * KV2_LD], xs[64 * KV2_LD], ydo[64 * KV2_LD], yqu[64 * KV2_LD];
    const int b = blockIdx.z, h = blockIdx.y, jt = (int)blockIdx.x / isplit, part = (int)blockIdx.x - jt * isplit, j0 = jt * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int jhalf = wave & 1, dhalf = wave >> 1;
    const int D = H * Dh;
    const long long row_stride = 3LL * D;
    const T *q_base = qkv + ((long long)b * Tn) * row_stride + (long long)h * 3 * Dh;
    const T *do_base = dout + ((long long)b * Tn) * D + (long long)h * Dh;
    const T *pd_base = pd + (((long long)b * H + h) * Tn) * Tp + j0;
    const T *ds_base = ds + (((long long)b * H + h) * Tn) * Tp + j0;
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    const int jk = j0 + 32 * jhalf + r;                    // this lane's key
    const bool k_live = jk < len;
    const int grp = lane >> 4, mhalf = grp & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int c = (tid & 7) * 8, rr0 = tid >> 3;           // staging: 8 column chunks x 32 rows per pass, 2 passes
    const bool fast_d = (Dh % 8) == 0;
    float bu8[8], bv8[8];     // (raw pieces: the dims at and beyond Dh are masked where the values are used)
    if (fast_d) { load8_raw<float, true>(bias_u + h * Dh, c, Dh, bu8); load8_raw<float, true>(bias_v + h * Dh, c, Dh, bv8); }
    else { load8_raw<float, false>(bias_u + h * Dh, c, Dh, bu8); load8_raw<float, false>(bias_v + h * Dh, c, Dh, bv8); }
    f32x16 dk = {0}, dv = {0};
    // queries that can reach these keys: all of them, or (causal) those at or after the first key of the workgroup
    const int i_begin = causal ? (((j0 / causal) * causal) / 64) * 64 : 0;
    float xpv[2][8], xsv[2][8], dov[2][8], qv8[2][8];
    auto request_t = [&](int i0n, auto fast_tag) {   // always-issued clamped loads, all 8 in flight together; masked at the LDS store
        constexpr bool F = decltype(fast_tag)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int ic = min(i0n + rr0 + 32 * it, Tn - 1);
            ld8(pd_base + (long long)ic * Tp + c, xpv[it]);
            ld8(ds_base + (long long)ic * Tp + c, xsv[it]);
            load8_raw<T, F>(do_base + (long long)ic * D, c, Dh, dov[it]);
            load8_raw<T, F>(q_base + (long long)ic * row_stride, c, Dh, qv8[it]);
        }
    };
    auto request = [&](int i0n) {
        if (fast_d) request_t(i0n, std::true_type{});
        else request_t(i0n, std::false_type{});
    };
    const int i_first = i_begin + 64 * part, i_step = 64 * isplit;
    if (i_first < Tn) request(i_first);
    for (int i0 = i_first; i0 < Tn; i0 += i_step) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rr = rr0 + 32 * it, i = i0 + rr;
            const bool live = i < Tn;
            float a[8], cv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = live && c + e < Dh;
                a[e] = ok ? qv8[it][e] + bu8[e] : 0.f;
                cv[e] = ok ? qv8[it][e] + bv8[e] : 0.f;
                if (!live) { xpv[it][e] = 0.f; xsv[it][e] = 0.f; }
                if (!ok) dov[it][e] = 0.f;
            }
            st8(xp + rr * KV2_LD + c, xpv[it]);
            st8(xs + rr * KV2_LD + c, xsv[it]);
            st8(ydo + rr * KV2_LD + c, dov[it]);
            st8(yqu + rr * KV2_LD + c, a);
            // (Q + v) rows in the [H, B*T, Dh] layout of the d(pk) product: written once, by the workgroup of the first key block
            if (qv_out && jt == 0 && live && c < Dh) {
                T *qo = qv_out + (((long long)h * gridDim.z + b) * Tn + i) * Dh + c;
                if (fast_d) st8(qo, cv);
                else
                    for (int e = 0; e < 8 && c + e < Dh; ++e) st1(qo + e, cv[e]);
            }
        }
        if (i0 + i_step < Tn) request(i0 + i_step);     // next chunk in flight during this chunk's MFMAs
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {           // 16 queries per MFMA step
            if (i0 + 16 * s >= Tn) break;
            const int ko = (16 * s + 4 * hh + q4) * KV2_LD + 16 * mhalf + 4 * p4;
            auto frag = [&](const bf16_t *tile, int col0) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + ko + col0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + ko + col0 + 8 * KV2_LD));
                bf16x8 f;
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                return f;
            };
            const bf16x8 a_do = frag(ydo, 32 * dhalf), a_qu = frag(yqu, 32 * dhalf);
            const bf16x8 b_p = frag(xp, 32 * jhalf), b_s = frag(xs, 32 * jhalf);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_do, b_p, dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_qu, b_s, dk, 0, 0, 0);
        }
    }
    if (isplit > 1) {   // fp32 share of this part: [dK 64 | dV 64] per key (rows of 128 floats; dims beyond Dh hold zeros of the zero-padded operands)
        if (jk < Tn) {
            float *pp = kv_part + ((((long long)part * gridDim.z + b) * H + h) * Tn + jk) * 128;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * dhalf + 8 * q + 4 * hh;
                *reinterpret_cast<float4 *>(pp + d) = make_float4(dk[4 * q], dk[4 * q + 1], dk[4 * q + 2], dk[4 * q + 3]);
                *reinterpret_cast<float4 *>(pp + 64 + d) = make_float4(dv[4 * q], dv[4 * q + 1], dv[4 * q + 2], dv[4 * q + 3]);
            }
        }
        return;
    }
    if (jk < Tn) {   // accumulators: rows = head dims 32*dhalf + (g&3) + 8(g>>2) + 4hh, column = this lane's key; masked keys get zeros
        T *dkp = dqkv + ((long long)b * Tn + jk) * row_stride + (long long)h * 3 * Dh + Dh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * dhalf + 8 * q + 4 * hh;
            float kk[4], vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { kk[e] = k_live ? dk[4 * q + e] : 0.f; vv[e] = k_live ? dv[4 * q + e] : 0.f; }
            if (d + 4 <= Dh && (Dh % 4) == 0) {
                st4(dkp + d, kk[0], kk[1], kk[2], kk[3]);
                st4(dkp + Dh + d, vv[0], vv[1], vv[2], vv[3]);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (d + e < Dh) { st1(dkp + d + e, kk[e]); st1(dkp + Dh + d + e, vv[e]); }
            }
        }
    }
}

// dK, dV of key (b, h, j) = sum over the parts of relpos_attn_bwd_kv2_kernel's fp32 shares, in part order; masked keys get zeros. 32 threads per key.
template <typename T>
__global__ __launch_bounds__(256) void relpos_attn_kv_merge_kernel(const float *__restrict__ kv_part, const int32_t *__restrict__ key_lens, T *__restrict__ dqkv,
                                                                   int B, int Tn, int H, int Dh, int isplit) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, key = gid >> 5;
    const int d4 = (int)(gid & 31) * 4;      // 0 .. 124: dK dims 0 .. 63, then dV dims 0 .. 63
    if (key >= (long long)B * H * Tn) return;
    const int j = (int)(key % Tn), h = (int)((key / Tn) % H), b = (int)(key / ((long long)Tn * H));
    const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < isplit; ++p) {
        const float4 v = *reinterpret_cast<const float4 *>(kv_part + (((long long)p * B * H * Tn) + key) * 128 + d4);
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
    if (j >= len) a[0] = a[1] = a[2] = a[3] = 0.f;
    const int which = d4 >> 6, d = d4 & 63;      // 0: dK, 1: dV
    T *dst = dqkv + ((long long)b * Tn + j) * (3LL * H * Dh) + (long long)h * 3 * Dh + Dh + which * Dh + d;
    if (d + 4 <= Dh && (Dh % 4) == 0) st4(dst, a[0], a[1], a[2], a[3]);
    else
        for (int e = 0; e < 4; ++e)
            if (d + e < Dh) st1(dst + e, a[e]);
}

// d(pk)[r][h*Dh + d] = sum_b sum_i dS[b][h][i][j = r + i - (T-1)] * (q + v)[b][i][h][d]  - the gradient of the projected positional
// table, straight from the materialised dS (the first version shifted dS back onto an (r, i) grid in HBM - 32 MB - and ran a library
// batched GEMM over it whose K dimension was half zeros: 20 + 31 + 5 us per layer). workgroup = (64 band rows, head, group of
// utterances); per utterance and block of 64 queries that can reach those rows: the 64 x 136 rectangle of dS goes to LDS as
// 16-byte pieces, the skew happens on the way into the A tile (A[rl][i] = dS[i][r0 + rl + i - (T-1)], 2-byte LDS reads, 16-byte
// writes), (q+v) rows are the B tile (k-major: transposing fragment reads), 4 MFMAs per wave; blocks of queries that cannot reach
// the rows are skipped (half of them). Partial sums per utterance group, summed by dpk_reduce_kernel in a fixed order.
#define SH_LD 136
#define DPK_LD 72
template <typename T>
__device__ __forceinline__ void dpk_body(const T *__restrict__ ds, const T *__restrict__ qv /*[H][B*T][Dh]*/,
                                         const int32_t *__restrict__ key_lens, float *__restrict__ part /*[G][R][H*64]*/,
                                         int Bn, int Tn, int Tp, int H, int Dh, int causal, int bgroup, int isplit, int i_span,
                                         int bx, int by, int bz) {
    // (bx, by, bz) = the block index of the one-job launch; isplit > 1 (long sequences, few utterances): bz = utterance group * isplit +
    // query range; a workgroup walks the query blocks of [ipart * i_span, (ipart + 1) * i_span) only - the band rows around r = T-1 are
    // reached by every query block
    __shared__ __attribute__((aligned(16))) T raw[64 * SH_LD];
    __shared__ __attribute__((aligned(16))) bf16_t a_tile[64 * DPK_LD], b_tile[64 * DPK_LD];
    const int r0 = bx * 64, h = by, grp = bz / isplit, ipart = bz % isplit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int rblk = wave & 1, dblk = wave >> 1;
    const int R = 2 * Tn - 1;
    const int grp4 = lane >> 4, mhalf = grp4 & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    constexpr int VE = 16 / (int)sizeof(T), NG = SH_LD / VE;
    f32x16 acc = {0};
    const int b_end = min(Bn, (grp + 1) * bgroup);
    const int i_lo = ipart * i_span, i_hi = min(Tn, i_lo + i_span);
    // band rows whose every (query, key) pair lies beyond the causal limit (j - i = r - (T-1) > chunk - 1) only ever see zeros
    const bool dead = causal && r0 - (Tn - 1) > max(causal, 1) - 1;
    // (a version that requested the next pair's global loads before building this pair's tiles - LDS-only barriers in between -
    // measured 5 % slower than this plain loop: two to four workgroups share a CU and cover each other's round trips)
    for (int b = grp * bgroup; b < b_end && !dead; ++b) {
        const int len = key_lens ? min(max(key_lens[b], 1), Tn) : Tn;
        const T *src = ds + (((long long)b * H + h) * Tn) * Tp;
        const T *qrow = qv + ((long long)h * Bn + b) * Tn * Dh;
        for (int i0 = i_lo; i0 < i_hi; i0 += 64) {
            const int jlo = r0 + i0 - (Tn - 1);                  // key of (rl = 0, il = 0); keys jlo .. jlo + 126 are touched
            if (jlo + 126 < 0 || jlo >= len) continue;           // no query of this block reaches these band rows (workgroup-uniform)
            const int jal = (jlo >= 0 ? jlo : jlo - 7) / 8 * 8, off = jlo - jal;
            __syncthreads();                                     // previous tiles consumed
            for (int e = tid; e < 64 * NG; e += 256) {           // always-issued clamped loads; validity decided below
                const int il = e / NG, gq = e % NG, j = jal + gq * VE;
                *reinterpret_cast<uint4 *>(raw + il * SH_LD + gq * VE) =
                    *reinterpret_cast<const uint4 *>(src + (long long)min(i0 + il, Tn - 1) * Tp + min(max(j, 0), Tp - VE));
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {                     // (q + v) rows i0 .. i0+63: 64 x 8 pieces of 8 dims
                const int e = tid + 256 * it, il = e >> 3, c = (e & 7) * 8;
                float v8[8];
                if ((Dh % 8) == 0) load8_raw<T, true>(qrow + (long long)min(i0 + il, Tn - 1) * Dh, c, Dh, v8);     // (both pieces of the
                else load8_raw<T, false>(qrow + (long long)min(i0 + il, Tn - 1) * Dh, c, Dh, v8);                  //  thread in flight together)
#pragma unroll
                for (int q = 0; q < 8; ++q) v8[q] = (c + q < Dh) ? v8[q] : 0.f;
                if (i0 + il >= Tn) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v8[q] = 0.f;
                }
                st8(b_tile + il * DPK_LD + c, v8);
            }
            __syncthreads();
            {   // A tile: thread = (band row rl, 16 consecutive queries)
                // All 16 reads are issued together and masked afterwards by a bit mask: written as `valid ? raw[..] : 0` with the four-term
                // validity test, each read sat in its own exec-masked block behind ~50 instructions of branches and was waited for on the
                // spot - 16 serialized LDS round trips per tile for 4 MFMAs. Without a look-ahead mask the valid elements of a thread are a
                // contiguous range of q (j = jlo + rl + il0 + q in [0, len), i0 + il0 + q < Tn).
                const int rl = tid >> 2, il0 = (tid & 3) * 16;
                float v16[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v16[q] = (float)raw[(il0 + q) * SH_LD + off + rl + il0 + q];     // always inside the 64 x 136 rectangle
                pin_all(v16);
                const int jq0 = jlo + rl + il0;                                                   // key of q = 0
                const int q_lo = min(max(-jq0, 0), 16), q_hi = min(max(min(len - jq0, Tn - i0 - il0), 0), 16);
                unsigned vm = q_hi > q_lo ? (0xffffu >> (16 - q_hi)) & (0xffffu << q_lo) : 0u;
                if (causal) {       // workgroup-uniform
#pragma unroll
                    for (int q = 0; q < 16; ++q) vm &= ~((jq0 + q > causal_limit(i0 + il0 + q, causal) ? 1u : 0u) << q);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) v16[q] = __uint_as_float(__float_as_uint(v16[q]) & (unsigned)((int)(vm << (31 - q)) >> 31));
                st8(a_tile + rl * DPK_LD + il0, *reinterpret_cast<float(*)[8]>(&v16[0]));
                st8(a_tile + rl * DPK_LD + il0 + 8, *reinterpret_cast<float(*)[8]>(&v16[8]));
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(a_tile + (32 * rblk + r) * DPK_LD + 16 * s + 8 * hh);
                const bf16_t *bp = b_tile + (16 * s + 8 * hh + q4) * DPK_LD + 32 * dblk + 16 * mhalf + 4 * p4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(bp + 4 * DPK_LD));
                bf16x8 bfr;
                bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3]; bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
            }
        }
    }
    // accumulator: rows = band rows 32*rblk + (g&3) + 8(g>>2) + 4hh, column = head dim 32*dblk + r
    float *pw = part + ((long long)bz * R) * (H * 64) + h * 64 + 32 * dblk + r;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int rg = r0 + 32 * rblk + (g & 3) + 8 * (g >> 2) + 4 * hh;
        if (rg < R) pw[(long long)rg * (H * 64)] = acc[g];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relpos_dpk_kernel(const T *__restrict__ ds, const T *__restrict__ qv, const int32_t *__restrict__ key_lens,
                                                         float *__restrict__ part, int Bn, int Tn, int Tp, int H, int Dh, int causal, int bgroup,
                                                         int isplit, int i_span) {
    dpk_body<T>(ds, qv, key_lens, part, Bn, Tn, Tp, H, Dh, causal, bgroup, isplit, i_span, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dpk[r][h*Dh + d] = sum over the utterance groups of part[g][r][h*64 + d], written in the io dtype (blocks `blk` of `nblk` of one job)
template <typename T>
__device__ __forceinline__ void dpk_reduce_body(const float *__restrict__ part, T *__restrict__ dpk, int R, int H, int Dh, int G, int blk, int nblk) {
    const long long n = (long long)R * H * 64;
    for (long long e = blk * 256LL + threadIdx.x; e < n; e += (long long)nblk * 256) {
        const int c = (int)(e % (H * 64)), d = c & 63, hq = c >> 6;
        const long long rr = e / (H * 64);
        if (d >= Dh) continue;
        float sum = 0.f;
        for (int g = 0; g < G; ++g) sum += part[(long long)g * n + e];
        st1(dpk + rr * (H * Dh) + hq * Dh + d, sum);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dpk_reduce_kernel(const float *__restrict__ part, T *__restrict__ dpk, int R, int H, int Dh, int G) {
    dpk_reduce_body<T>(part, dpk, R, H, Dh, G, blockIdx.x, gridDim.x);
}

// ---- deferred d(pk): the pass feeds nothing in backward but the weight gradient of linear_pos (pos_embs carries no gradient), so the
// training step only QUEUES it (tsasr_relpos_dpk_defer) and runs the passes of all layers as ONE launch (+ one for the sums of the
// partials) right before the grouped weight-gradient launch that consumes them: 12 + 6 pairs of latency-bound launches (25 + 10 us and
// 13 + 11 us per layer at configs[1], two dependent launch gaps each) leave the critical path of the encoder's backward, and the one
// launch fills the chip. Same blocks, same arithmetic, same order of sums as the per-layer launches: bit-identical d(pk).
struct DpkJob {
    const void *ds, *qv;
    const int32_t *key_lens;
    float *part;
    void *dpk;
    int Bn, Tn, Tp, H, Dh, causal, bgroup, isplit, i_span, G, nbx, io_dtype;
    int tile0, ntiles, rtile0, rtiles;      // first block / block count of this job in the grouped pass and in the grouped reduction
};

__device__ __forceinline__ int dpk_find_job(const DpkJob *__restrict__ jobs, int njobs, int bid, bool reduce) {
    int k = 0;
    while (k + 1 < njobs && (reduce ? jobs[k + 1].rtile0 : jobs[k + 1].tile0) <= bid) ++k;
    return __builtin_amdgcn_readfirstlane(k);
}

template <typename T>
__global__ __launch_bounds__(256) void relpos_dpk_group_kernel(const DpkJob *__restrict__ jobs, int njobs) {
    const DpkJob j = jobs[dpk_find_job(jobs, njobs, blockIdx.x, false)];
    const int lid = blockIdx.x - j.tile0;
    dpk_body<T>((const T *)j.ds, (const T *)j.qv, j.key_lens, j.part, j.Bn, j.Tn, j.Tp, j.H, j.Dh, j.causal, j.bgroup, j.isplit, j.i_span,
                lid % j.nbx, (lid / j.nbx) % j.H, lid / (j.nbx * j.H));
}

template <typename T>
__global__ __launch_bounds__(256) void dpk_reduce_group_kernel(const DpkJob *__restrict__ jobs, int njobs) {
    const DpkJob j = jobs[dpk_find_job(jobs, njobs, blockIdx.x, true)];
    dpk_reduce_body<T>(j.part, (T *)j.dpk, 2 * j.Tn - 1, j.H, j.Dh, j.G, blockIdx.x - j.rtile0, j.rtiles);
}

extern "C" size_t tsasr_relpos_attn_lds_bytes(void);
// csrc/attention_short.hip
extern "C" int tsasr_attn_short_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, void *out,
                                    float *lse, int B, int T, int H, float scale, int causal, float pdrop, unsigned long long seed,
                                    const unsigned long long *seed_dev, void *keepbits, hipStream_t st);
extern "C" int tsasr_attn_chunk_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, int B, int T, int H,
                                    float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int nparts,
                                    float *part_o, float *part_ml, hipStream_t st);
extern "C" int tsasr_attn_short_bwd_q(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *out,
                                      const void *dout, const float *lse, void *dqkv, void *pd, void *ds, float *slab, int Tp, int B, int T, int H,
                                      float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev,
                                      const void *keepbits, hipStream_t st);
extern "C" int tsasr_attn_short_bwd_kv(const void *qkv, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *dout, const void *pd,
                                       const void *ds, void *dqkv, void *qv_out, int B, int T, int Tp, int H, int causal, hipStream_t st);
constexpr int AT_CHUNK_KEYS = 256;
static bool attn_chunk_enabled() {      // TSASR_ATTN_CHUNK=0: the streaming forward for long sequences (A/B runs)
    static const bool on = [] { const char *e = getenv("TSASR_ATTN_CHUNK"); return !e || e[0] != '0'; }();
    return on;
}
static bool attn_kv_split_enabled() {      // TSASR_ATTN_KV_SPLIT=0: one workgroup per key tile in the streaming key-major pass (A/B)
    static const bool on = [] { const char *e = getenv("TSASR_ATTN_KV_SPLIT"); return !e || e[0] != '0'; }();
    return on;
}
static bool attn_zero_band_enabled() {      // TSASR_ATTN_ZERO_BAND=0: clear the whole P_d / dS matrices under a look-ahead mask (A/B, round 4's form)
    static const bool on = [] { const char *e = getenv("TSASR_ATTN_ZERO_BAND"); return !e || e[0] != '0'; }();
    return on;
}
static int attn_short_version() {   // TSASR_ATTN_SHORT = 1: round 3's short-sequence forward (A/B); default 2
    static const int v = [] { const char *e = getenv("TSASR_ATTN_SHORT"); return e ? atoi(e) : 2; }();
    return v;
}

// ---- key parts for long sequences in small batches (forward and query-major backward) -------------------------------------------
// One workgroup per (utterance, head, 128 queries) leaves the chip under-filled when B * H * T/128 < CUs, and under a causal mask the
// blocks need 2 .. T/64 key tiles each: the keys are then cut into parts of `per` tiles, `per` the smallest value (>= 8) for which the
// non-empty workgroups still fit one per CU (two per CU share its LDS / VALU throughput: T' = 4000, B = 1 measured 131 us per layer
// with 24-tile parts = 240 workgroups, 136 - 193 us with 4 .. 16-tile parts, 297 us unsplit).
constexpr int AT_MIN_PART = 8;
static int host_causal_limit(int i, int causal) { return causal <= 1 ? i : (i / causal + 1) * causal - 1; }
static int attn_ksplit_env() {   // TSASR_ATTN_KSPLIT = 0: never split; n > 0: parts of n key tiles whatever the shape (A/B)
    static const int v = -1;
    return v;
}
static int attn_max_parts(int T) {
    const int f = attn_ksplit_env();
    return std::max(1, cdiv(cdiv(T, AT_KT), f > 0 ? std::min(f, AT_MIN_PART) : AT_MIN_PART));
}
static int attn_key_parts(int B, int T, int H, int causal, int *part_keys) {
    const int forced = attn_ksplit_env();
    static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    const int tiles = cdiv(T, AT_KT), nqb = cdiv(T, AT_QB);
    *part_keys = 0;
    if (forced > 0) {
        if (forced >= tiles) return 1;
        *part_keys = forced * AT_KT;
        return cdiv(tiles, forced);
    }
    if (forced == 0 || tiles <= AT_MIN_PART || (long long)B * H * nqb >= cus) return 1;
    int per = AT_MIN_PART;
    if (forced <= 0)
        for (; per < tiles; ++per) {
            long long n = 0;
            for (int qb = 0; qb < nqb; ++qb) {
                const int je = causal ? std::min(T, host_causal_limit(qb * AT_QB + AT_QB - 1, causal) + 1) : T;
                n += cdiv(cdiv(je, AT_KT), per);
            }
            if (n * B * H <= cus) break;
        }
    if (per >= tiles) return 1;
    *part_keys = per * AT_KT;
    return cdiv(tiles, per);
}

static size_t attn_slab_bytes(int B, int T, int H) { return align_up((size_t)B * cdiv(T, AT_QB) * attn_max_parts(T) * H * AT_NW * 128 * sizeof(float), 256); }
static size_t attn_dq_part_bytes(int B, int T, int H) { return attn_max_parts(T) > 1 ? align_up((size_t)B * H * T * attn_max_parts(T) * AT_DP * sizeof(float), 256) : 0; }
static int attn_tp(int T) { return cdiv(T, 64) * 64; }
// d(pk): query ranges per workgroup when (band-row blocks x heads x utterance groups) alone would be few, long walks
static int attn_dpk_isplit(int B, int T, int H, int causal, int G) {
    static const int forced = 0;
    const int nib = cdiv(T, 64);
    if (forced > 0) return std::min(forced, nib);
    const long long live = (long long)cdiv(causal ? T + std::max(causal, 1) : 2 * T - 1, 64) * H * G;   // causal: band rows beyond T-1+chunk are dead
    if (live >= 1024 || nib < 16) return 1;
    return (int)std::min<long long>(cdiv(1024, (int)live), nib / 8);
}
static int attn_dpk_max_isplit(int T) { return std::max(1, std::min(cdiv(T, 64) / 8, 16)); }
// utterances per d(pk) workgroup: ~256 workgroups per layer. The training step runs the passes of 12 (6) layers as ONE grouped launch, which
// fills the chip anyway; fewer, longer workgroups write a quarter of the partial planes and leave the reduction a quarter to add
// (configs[1], ms per step at 1024 / 512 / 256 / 128 workgroups per layer: 11.88 / 11.83 / 11.80 / 11.87). A stand-alone launch (one
// layer, no gradient arena) under-fills the chip with it - the price of keeping both forms bit-identical.
static int attn_bgroup(int B, int T) {
    static const int dpk_wgs = 256;
    const int want = std::max(1, dpk_wgs / (4 * cdiv(2 * T - 1, 64)));
    return std::max(1, cdiv(B, std::min(B, want)));
}
static size_t attn_qv_bytes(int B, int T, int H) { return align_up((size_t)B * T * H * AT_DP * sizeof(float), 256); }
static size_t attn_part_bytes(int B, int T, int H) {
    return align_up((size_t)cdiv(B, attn_bgroup(B, T)) * attn_dpk_max_isplit(T) * (2 * T - 1) * H * 64 * sizeof(float), 256);
}

static std::vector<DpkJob> g_dpk_jobs;
static int g_dpk_defer = 0;
static void *g_attn_keepbits = nullptr;     // tsasr_relpos_attn_keepbits: the next forward writes / the next backward reads them

// Streaming backward under a look-ahead mask: the query-major pass skips the 32-key blocks that lie beyond the limit of a whole 32-query wave,
// and the key-major pass (relpos_attn_bwd_kv2_kernel) reads, for a tile of 64 keys, every query from the tile's first reachable 64-query chunk
// on - without a mask. What it can meet unwritten is therefore a band to the right of the diagonal: row i, columns [32 floor(i / 32),
// i + chunk + 192). Clearing that band of both matrices (<= ~0.3 KB per row and matrix) replaces clearing both whole matrices (2 x 129 MB per
// layer at T' = 4000: 35 us). The d(pk) pass masks what it reads by the look-ahead rule itself.
template <typename T>
__global__ __launch_bounds__(256) void attn_zero_band_kernel(T *__restrict__ pd, T *__restrict__ ds, long long rows, int Tn, int Tp, int chunk) {
    constexpr int VE = 16 / (int)sizeof(T);
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % Tn), lane = threadIdx.x & 63;
    const int lo = (i / 32) * 32, hi = min(Tp, i + chunk + 192);      // (lo is a multiple of 32 elements: 16-byte aligned for both types; Tp % 8 == 0)
    for (int c = lo + lane * VE; c < hi; c += 64 * VE) {
        *reinterpret_cast<uint4 *>(pd + row * Tp + c) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(ds + row * Tp + c) = make_uint4(0u, 0u, 0u, 0u);
    }
}

__global__ void attn_zero_kernel(uint4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

template <typename T>
static void launch_attn_bwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens, const void *out,
                            const void *dout, const float *lse, void *dqkv, void *dpk, int B, int Tn, int H, int Dh, float scale,
                            int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, char *workspace, hipStream_t st) {
    const int Tp = attn_tp(Tn);
    float *slab = (float *)workspace;
    const size_t mat = align_up((size_t)B * H * Tn * Tp * sizeof(float), 256);
    T *pd = (T *)(workspace + attn_slab_bytes(B, Tn, H)), *ds = (T *)(workspace + attn_slab_bytes(B, Tn, H) + mat);
    T *qv = (T *)(workspace + attn_slab_bytes(B, Tn, H) + 2 * mat);
    float *part = (float *)(workspace + attn_slab_bytes(B, Tn, H) + 2 * mat + attn_qv_bytes(B, Tn, H));
    float *dq_part = (float *)(workspace + attn_slab_bytes(B, Tn, H) + 2 * mat + attn_qv_bytes(B, Tn, H) + attn_part_bytes(B, Tn, H));
    const int bg = attn_bgroup(B, Tn), G = cdiv(B, bg), R = 2 * Tn - 1;
    int part_keys = 0;
    const int nparts = attn_key_parts(B, Tn, H, causal, &part_keys);
    const int isplit = std::min(attn_dpk_isplit(B, Tn, H, causal, G), attn_dpk_max_isplit(Tn)), i_span = cdiv(cdiv(Tn, 64), isplit) * 64;
    const bool short_path = sizeof(T) == 2 && Dh == 64 && Tn <= 256 && Tn >= 2 && nparts == 1 && attn_short_version() >= 2;
    if (causal && (short_path || Tp % (16 / (int)sizeof(T)) != 0 || !attn_zero_band_enabled())) {
        // key blocks in the future of a whole query wave are skipped by bwd_q: their entries must read as zero (the short key-major pass streams
        // every query chunk past its keys: the whole matrices)
        const size_t used = align_up((size_t)B * H * Tn * Tp * sizeof(T), 16) / 16;   // `mat` is sized for fp32 io
        attn_zero_kernel<<<1024, 256, 0, st>>>((uint4 *)pd, used);
        attn_zero_kernel<<<1024, 256, 0, st>>>((uint4 *)ds, used);
    } else if (causal) {
        const long long rows = (long long)B * H * Tn;
        attn_zero_band_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(pd, ds, rows, Tn, Tp, std::max(causal, 1));
    }
    const size_t lds_q = tsasr_relpos_attn_lds_bytes();
    const unsigned short *kbq = (Tn <= 256 && sizeof(T) == 2) ? (const unsigned short *)g_attn_keepbits : nullptr;    // one-shot
    g_attn_keepbits = nullptr;
    // A/B knob TSASR_ATTN_BWD_KG=2: two key groups per workgroup (8 waves, two per SIMD). Measured at T' = 250, B = 32 (one workgroup per
    // CU): 13.07 vs 13.06 ms per step with the tile prefetch dropped to fit 256 registers, 13.61 with it (76 spilled VGPRs) - the kernel
    // is bound by the CU's VALU / LDS instruction throughput, not by latency a second wave could cover. Off.
    static const int kg_env = 1;
    const bool two = kg_env == 2 && Tn > AT_KT;
    if (sizeof(T) == 2 && Dh == 64 && Tn <= 256 && Tn >= 2 && nparts == 1 && attn_short_version() >= 2) {
        tsasr_attn_short_bwd_q(qkv, pk, bias_u, bias_v, key_lens, out, dout, lse, dqkv, pd, ds, slab, Tp, B, Tn, H, scale, causal, pdrop, seed, seed_dev, kbq, st);
    } else if (two) {
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_q_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_q));
        relpos_attn_bwd_q_kernel<T, 2><<<dim3(cdiv(Tn, AT_QB) * nparts, H, B), 2 * AT_TH, 2 * lds_q, st>>>(
            (const T *)qkv, (const T *)pk, bias_u, bias_v, key_lens, (const T *)out, (const T *)dout, lse, (T *)dqkv, pd, ds, slab, Tp, B, Tn, H, Dh, scale,
            causal, pdrop, seed, seed_dev, nparts, part_keys, dq_part, kbq);
    } else {
        (void)hipFuncSetAttribute((const void *)relpos_attn_bwd_q_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        relpos_attn_bwd_q_kernel<T, 1><<<dim3(cdiv(Tn, AT_QB) * nparts, H, B), AT_TH, lds_q, st>>>(
            (const T *)qkv, (const T *)pk, bias_u, bias_v, key_lens, (const T *)out, (const T *)dout, lse, (T *)dqkv, pd, ds, slab, Tp, B, Tn, H, Dh, scale,
            causal, pdrop, seed, seed_dev, nparts, part_keys, dq_part, kbq);
    }
    if (nparts > 1)
        relpos_attn_dq_merge_kernel<T><<<(unsigned)(((long long)B * H * Tn * 16 + 255) / 256), 256, 0, st>>>(dq_part, key_lens, (T *)dqkv, B, Tn, H, Dh, causal,
                                                                                                          nparts, part_keys);
    if (sizeof(T) == 2 && Dh == 64 && Tn <= 256 && Tn >= 2 && nparts == 1 && attn_short_version() >= 2)
        tsasr_attn_short_bwd_kv(qkv, bias_u, bias_v, key_lens, dout, pd, ds, dqkv, qv, B, Tn, Tp, H, causal, st);
    else {
        // few long utterances: four parts per key tile (interleaved query chunks: equal shares whatever the look-ahead mask), merged in part order;
        // the shares live where the query-major pass's dQ shares were (already merged)
        const int ks = (nparts > 1 && attn_kv_split_enabled() && attn_dq_part_bytes(B, Tn, H) >= (size_t)4 * B * H * Tn * 128 * sizeof(float)) ? 4 : 1;
        relpos_attn_bwd_kv2_kernel<T><<<dim3(cdiv(Tn, 64) * ks, H, B), 256, 0, st>>>((const T *)qkv, bias_u, bias_v, key_lens, (const T *)dout, pd, ds, (T *)dqkv,
                                                                                   qv, Tn, Tp, H, Dh, causal, ks, dq_part);
        if (ks > 1)
            relpos_attn_kv_merge_kernel<T><<<(unsigned)(((long long)B * H * Tn * 32 + 255) / 256), 256, 0, st>>>(dq_part, key_lens, (T *)dqkv, B, Tn, H, Dh, ks);
    }
    if (g_dpk_defer) {   // queued: both passes run in tsasr_relpos_dpk_flush (workspace, key_lens and dpk stay alive until then)
        DpkJob j{ds, qv, key_lens, part, dpk, B, Tn, Tp, H, Dh, causal, bg, isplit, i_span, G * isplit, cdiv(R, 64),
                 sizeof(T) == 2 ? TSASR_BF16 : TSASR_F32, 0, cdiv(R, 64) * H * G * isplit, 0, std::min(1024, cdiv(R * H * 64, 256))};
        g_dpk_jobs.push_back(j);
        return;
    }
    relpos_dpk_kernel<T><<<dim3(cdiv(R, 64), H, G * isplit), 256, 0, st>>>(ds, qv, key_lens, part, B, Tn, Tp, H, Dh, causal, bg, isplit, i_span);
    dpk_reduce_kernel<T><<<std::min(1024, cdiv(R * H * 64, 256)), 256, 0, st>>>(part, (T *)dpk, R, H, Dh, G * isplit);
}

extern "C" {

size_t tsasr_relpos_attn_lds_bytes(void) {
    return (size_t)(2 * AT_KT + AT_BAND) * AT_LD * sizeof(bf16_t) + (size_t)AT_NW * 64 * 32 * sizeof(float);
}

size_t tsasr_relpos_attn_fwd_workspace_bytes(int B, int T, int H) {   // sized for the finest split any mask may choose
    int pk = 0;
    const int np = std::max(attn_key_parts(B, T, H, 0, &pk), attn_key_parts(B, T, H, 1, &pk)) > 1 ? std::max(attn_max_parts(T), cdiv(T, AT_CHUNK_KEYS)) : 1;
    return np > 1 ? align_up((size_t)B * H * T * np * (AT_DP + 2) * sizeof(float), 256) : 0;
}

/* out [B,T,H*Dh] = fused rel-pos attention; lse [B,H,T] fp32 (may be NULL) is kept for the backward.
 * Dh <= 64; dropout mask is a pure function of (seed, b, h, i, j). */
int tsasr_relpos_attn_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                          void *out, float *lse, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                          unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream) {
    return tsasr_relpos_attn_fwd_ws(qkv, pk, bias_u, bias_v, key_lens, out, lse, B, T, H, Dh, scale, causal, pdrop, seed, seed_dev, io_dtype,
                                    nullptr, 0, stream);
}

/* Same with a workspace of tsasr_relpos_attn_fwd_workspace_bytes(B, T, H) bytes: long sequences in small batches are then split along
 * the keys across workgroups (partial results merged by a second launch); without a workspace (NULL) nothing is split. */
int tsasr_relpos_attn_fwd_ws(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                             void *out, float *lse, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                             unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                             size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(qkv && pk && bias_u && bias_v && out, "tsasr_relpos_attn_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && H > 0 && Dh > 0 && Dh <= AT_DP, "tsasr_relpos_attn_fwd: head dim %d not supported (1..%d)", Dh, AT_DP);
    TSASR_CHECK_ARG(pdrop >= 0.f && pdrop < 1.f, "tsasr_relpos_attn_fwd: bad dropout %f", pdrop);
    const size_t lds = tsasr_relpos_attn_lds_bytes();
    dim3 grid(cdiv(T, AT_QB), H, B);
    hipStream_t st = (hipStream_t)stream;
    static const int use_short = 1;
    unsigned short *kb = (unsigned short *)g_attn_keepbits;     // one-shot (tsasr_relpos_attn_keepbits); only the short-sequence kernel writes them
    g_attn_keepbits = nullptr;
    if (use_short && io_dtype == TSASR_BF16 && Dh == 64 && T <= 256 && T >= 2 && attn_short_version() >= 2) {
        tsasr_attn_short_fwd(qkv, pk, bias_u, bias_v, key_lens, out, lse, B, T, H, scale, causal, pdrop, seed, seed_dev, kb, st);
        TSASR_CHECK_LAUNCH("tsasr_relpos_attn_fwd");
        return 0;
    }
    if (use_short && io_dtype == TSASR_BF16 && Dh == 64 && T <= 256 && T >= 2) {
        if (T > 128) {
            constexpr int LDSS = (2 * 256 + 128 + 256) * 128 + 8 * 4096;   // K, V (256 rows), band (384 rows), 8 fp16 G tiles
            (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_short_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
            relpos_attn_fwd_short_kernel<128><<<dim3(cdiv(T, 128), H, B), 512, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens,
                                                                                        (bf16_t *)out, lse, T, H, scale, causal, pdrop, seed, seed_dev, kb);
        } else {
            constexpr int LDSS = (2 * 128 + 64 + 128) * 128 + 8 * 4096;
            (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_short_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
            relpos_attn_fwd_short_kernel<64><<<dim3(cdiv(T, 64), H, B), 512, LDSS, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens,
                                                                                      (bf16_t *)out, lse, T, H, scale, causal, pdrop, seed, seed_dev, kb);
        }
        TSASR_CHECK_LAUNCH("tsasr_relpos_attn_fwd");
        return 0;
    }
    int part_keys = 0;
    int nparts = workspace ? attn_key_parts(B, T, H, causal, &part_keys) : 1;
    if (nparts > 1 && workspace_bytes < tsasr_relpos_attn_fwd_workspace_bytes(B, T, H)) nparts = 1;
    if (nparts > 1 && io_dtype == TSASR_BF16 && Dh == 64 && cdiv(T, AT_CHUNK_KEYS) <= 64 && attn_chunk_enabled()) {
        // long sequence in a small batch: the everything-in-LDS kernel of the short sequences, one workgroup per chunk of 256 keys
        // (csrc/attention_short.hip), merged like the streaming kernel's key parts
        const int np = cdiv(T, AT_CHUNK_KEYS);
        float *po = (float *)workspace, *pml = po + (size_t)B * H * T * np * AT_DP;
        tsasr_attn_chunk_fwd(qkv, pk, bias_u, bias_v, key_lens, B, T, H, scale, causal, pdrop, seed, seed_dev, np, po, pml, st);
        relpos_attn_merge_kernel<bf16_t><<<(unsigned)(((long long)B * H * T * 16 + 255) / 256), 256, 0, st>>>(po, pml, key_lens, (bf16_t *)out, lse, B, T, H, Dh, causal, np,
                                                                                                               AT_CHUNK_KEYS);
        TSASR_CHECK_LAUNCH("tsasr_relpos_attn_fwd");
        return 0;
    }
    float *part_o = (float *)workspace, *part_ml = nparts > 1 ? part_o + (size_t)B * H * T * nparts * AT_DP : nullptr;
    if (nparts > 1) grid.x *= nparts;
    const unsigned mgrid = (unsigned)(((long long)B * H * T * 16 + 255) / 256);
    if (io_dtype == TSASR_F32) {
        (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        relpos_attn_fwd_kernel<float><<<grid, AT_TH, lds, st>>>((const float *)qkv, (const float *)pk, bias_u, bias_v, key_lens, (float *)out, lse, T, H, Dh, scale, causal, pdrop, seed, seed_dev, nparts, part_keys, part_o, part_ml);
        if (nparts > 1) relpos_attn_merge_kernel<float><<<mgrid, 256, 0, st>>>(part_o, part_ml, key_lens, (float *)out, lse, B, T, H, Dh, causal, nparts, part_keys);
    } else if (io_dtype == TSASR_BF16) {
        (void)hipFuncSetAttribute((const void *)relpos_attn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        relpos_attn_fwd_kernel<bf16_t><<<grid, AT_TH, lds, st>>>((const bf16_t *)qkv, (const bf16_t *)pk, bias_u, bias_v, key_lens, (bf16_t *)out, lse, T, H, Dh, scale, causal, pdrop, seed, seed_dev, nparts, part_keys, part_o, part_ml);
        if (nparts > 1) relpos_attn_merge_kernel<bf16_t><<<mgrid, 256, 0, st>>>(part_o, part_ml, key_lens, (bf16_t *)out, lse, B, T, H, Dh, causal, nparts, part_keys);
    } else {
        TSASR_CHECK_ARG(false, "tsasr_relpos_attn_fwd: bad io_dtype %d", io_dtype);
    }
    TSASR_CHECK_LAUNCH("tsasr_relpos_attn_fwd");
    return 0;
}

size_t tsasr_relpos_attn_bwd_workspace_bytes(int B, int T, int H) {
    // pos_bias partial sums + the two materialised [B,H,T,Tp] tensors (P_d, scale*dS; sized for fp32 io) + (q + v) rows + d(pk) partials
    // + fp32 dQ shares of the key parts (long sequences in small batches only)
    return attn_slab_bytes(B, T, H) + 2 * align_up((size_t)B * H * T * attn_tp(T) * sizeof(float), 256) + attn_qv_bytes(B, T, H) +
           attn_part_bytes(B, T, H) + attn_dq_part_bytes(B, T, H);
}

/* Backward of tsasr_relpos_attn_fwd. dqkv [B,T,H,3*Dh] (fully written), d_bias_u / d_bias_v fp32 [H*Dh] in the [H,Dh] reading of the
 * parameter storage, dpk [2T-1, H*Dh] (io_dtype, fully written) = gradient of the projected positional table pk. Five launches: the
 * query-major pass (recomputes the probabilities; dQ, bias partial sums, and P_d / scale*dS materialised in the workspace), the
 * key-major pass (dK, dV as plain contractions of those two tensors; also leaves the q + pos_bias_v rows), the d(pk) pass (skewed
 * gather of dS against those rows on MFMA, zero blocks skipped) and the sum of its per-utterance-group partials. */
int tsasr_relpos_attn_bwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                          const void *out, const void *dout, const float *lse, void *dqkv, void *dpk, float *d_bias_u,
                          float *d_bias_v, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                          unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace, size_t workspace_bytes,
                          void *stream) {
    TSASR_CHECK_ARG(qkv && pk && bias_u && bias_v && out && dout && lse && dqkv && dpk && d_bias_u && d_bias_v && workspace,
                    "tsasr_relpos_attn_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && H > 0 && Dh > 0 && Dh <= AT_DP, "tsasr_relpos_attn_bwd: head dim %d not supported", Dh);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_relpos_attn_bwd_workspace_bytes(B, T, H), "tsasr_relpos_attn_bwd: workspace too small");
    TSASR_CHECK_ARG((long long)B * H <= 65535, "tsasr_relpos_attn_bwd: B*H too large");
    int part_keys_ = 0;
    const int nqt = cdiv(T, AT_QB) * attn_key_parts(B, T, H, causal, &part_keys_);   // rows of pos_bias partial sums per utterance and wave
    hipStream_t st = (hipStream_t)stream;
    float *slab = (float *)workspace;
    if (io_dtype == TSASR_F32)
        launch_attn_bwd<float>(qkv, pk, bias_u, bias_v, key_lens, out, dout, lse, dqkv, dpk, B, T, H, Dh, scale, causal, pdrop, seed, seed_dev,
                               (char *)workspace, st);
    else if (io_dtype == TSASR_BF16)
        launch_attn_bwd<bf16_t>(qkv, pk, bias_u, bias_v, key_lens, out, dout, lse, dqkv, dpk, B, T, H, Dh, scale, causal, pdrop, seed, seed_dev,
                                (char *)workspace, st);
    else
        TSASR_CHECK_ARG(false, "tsasr_relpos_attn_bwd: bad io_dtype %d", io_dtype);
    // slab parts [(b, qtile, wave)] x row [h][u 64 | v 64] -> sum over the parts per (h, which, d): batched with the other
    // parameter-gradient reductions while tsasr_reduce_defer is on (csrc/reduce.hip)
    for (int h = 0; h < H; ++h) {
        tsasr_reduce_submit(slab + h * 128, d_bias_u + h * Dh, (long long)H * 128, AT_NW * B * nqt, Dh, 0, st);
        tsasr_reduce_submit(slab + h * 128 + 64, d_bias_v + h * Dh, (long long)H * 128, AT_NW * B * nqt, Dh, 0, st);
    }
    TSASR_CHECK_LAUNCH("tsasr_relpos_attn_bwd");
    return 0;
}

/* 1: tsasr_relpos_attn_bwd queues its d(pk) pass instead of launching it (the workspace, key_lens and dpk of every queued call must stay
 * alive and untouched until tsasr_relpos_dpk_flush); 0: launched inside the call (default). dpk is NOT written until the flush. */
int tsasr_relpos_dpk_defer(int on) {
    TSASR_CHECK_ARG(on || g_dpk_jobs.empty(), "tsasr_relpos_dpk_defer(0) with %d passes still queued: call tsasr_relpos_dpk_flush first", (int)g_dpk_jobs.size());
    g_dpk_defer = on ? 1 : 0;
    return 0;
}
int tsasr_relpos_dpk_pending(void) { return (int)g_dpk_jobs.size(); }
size_t tsasr_relpos_dpk_table_bytes(int max_jobs) { return (size_t)max_jobs * sizeof(DpkJob); }
void tsasr_relpos_dpk_discard(void) { g_dpk_jobs.clear(); g_dpk_defer = 0; }
/* One-shot side channel for the dropout keep-bits of the short-sequence kernels (bf16, Dh = 64, T <= 256; tsasr_relpos_attn_keepbits_bytes):
 * set before tsasr_relpos_attn_fwd*, the forward stores them there; set to the same buffer before tsasr_relpos_attn_bwd, the backward
 * reads them instead of hashing the mask again. NULL (or never called): both hash - same bits either way. */
void tsasr_relpos_attn_keepbits(void *bits) { g_attn_keepbits = bits; }
size_t tsasr_relpos_attn_keepbits_bytes(int B, int T, int H) { return (T >= 2 && T <= 256) ? (size_t)B * H * T * 16 * sizeof(unsigned short) : 0; }

/* Runs every queued d(pk) pass: ONE launch for the passes, one for the sums of their partials (per io dtype present). table_host: PINNED
 * host memory, table_dev: device memory, both >= tsasr_relpos_dpk_table_bytes(tsasr_relpos_dpk_pending()); copied host -> device on `stream`
 * except while it is being captured (then the caller uploads table_host after the capture: the protocol of tsasr_wgrad_flush). */
int tsasr_relpos_dpk_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream) {
    if (g_dpk_jobs.empty()) return 0;
    hipStream_t st = (hipStream_t)stream;
    std::stable_sort(g_dpk_jobs.begin(), g_dpk_jobs.end(), [](const DpkJob &a, const DpkJob &b) { return a.io_dtype < b.io_dtype; });
    const size_t need = g_dpk_jobs.size() * sizeof(DpkJob);
    TSASR_CHECK_ARG(table_host && table_dev && table_bytes >= need, "tsasr_relpos_dpk_flush: job table too small (%zu < %zu bytes)", table_bytes, need);
    struct Group { int first, count, tiles, rtiles, dtype; };
    std::vector<Group> groups;
    for (size_t i = 0; i < g_dpk_jobs.size(); ++i) {
        DpkJob &j = g_dpk_jobs[i];
        if (groups.empty() || groups.back().dtype != j.io_dtype) groups.push_back(Group{(int)i, 0, 0, 0, j.io_dtype});
        Group &g = groups.back();
        j.tile0 = g.tiles; j.rtile0 = g.rtiles;
        g.tiles += j.ntiles; g.rtiles += j.rtiles; g.count += 1;
    }
    memcpy(table_host, g_dpk_jobs.data(), need);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap == hipStreamCaptureStatusNone) {
        hipError_t e = hipMemcpyAsync(table_dev, table_host, need, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            tsasr_set_error("tsasr_relpos_dpk_flush: job table upload failed: %s", hipGetErrorString(e));
            return TSASR_E_LAUNCH;
        }
    }
    for (const Group &g : groups) {
        const DpkJob *tab = (const DpkJob *)table_dev + g.first;
        if (g.dtype == TSASR_BF16) {
            relpos_dpk_group_kernel<bf16_t><<<g.tiles, 256, 0, st>>>(tab, g.count);
            dpk_reduce_group_kernel<bf16_t><<<g.rtiles, 256, 0, st>>>(tab, g.count);
        } else {
            relpos_dpk_group_kernel<float><<<g.tiles, 256, 0, st>>>(tab, g.count);
            dpk_reduce_group_kernel<float><<<g.rtiles, 256, 0, st>>>(tab, g.count);
        }
    }
    g_dpk_jobs.clear();
    TSASR_CHECK_LAUNCH("tsasr_relpos_dpk_flush");
    return 0;
}

}  // extern "C"
