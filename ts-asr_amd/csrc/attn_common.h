// Helpers shared by the relative-position attention kernels (csrc/attention.hip: streaming kernels for any length;
// csrc/attention_short.hip: everything-resident kernels for T <= 256, bf16, Dh = 64).
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// `causal` argument of every kernel: 0 = no look-ahead mask; 1 = the reference's look-ahead mask (a frame sees keys j <= i:
// Transformer.py:890-914 via models/conformer.py:279-280); C > 1 = BUILD EXTENSION "chunk = C frames" (BASELINE.json configs[4]):
// block-causal, a frame sees its whole chunk of C frames and everything before it. Last key query i may attend:
__device__ __forceinline__ int causal_limit(int i, int causal) { return causal <= 1 ? i : (i / causal + 1) * causal - 1; }

// Opaque to the optimiser: the value must exist in a register HERE. Used on LDS reads whose only consumer sits behind a mask test:
// left alone, hipcc sinks each read into the branch that uses it - 16 guarded reads = 16 serialized LDS round trips per 32 x 32 score
// block (s_and_saveexec / ds_read / s_waitcnt lgkmcnt(0) each; "a guarded load is a serialized load").
__device__ __forceinline__ float pin(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
// The same for a whole batch of reads: ONE point where all of them must exist, so they are issued back to back and waited for once
__device__ __forceinline__ void pin_all(float (&a)[8]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}
__device__ __forceinline__ void pin_all(float (&a)[16]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                      "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
__device__ __forceinline__ void pin_all(unsigned (&a)[8]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

// four consecutive elements in one store (8 bytes of bf16, 16 of fp32)
__device__ __forceinline__ void st4(float *p, float a, float b, float c, float d) { *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ void st4(bf16_t *p, float a, float b, float c, float d) { *reinterpret_cast<uint2 *>(p) = make_uint2(pk_bf16(a, b), pk_bf16(c, d)); }

// LDS-DMA: 64 lanes x 16 bytes (one 1-KiB piece) / 64 lanes x 4 bytes from per-lane global addresses to lds_dst + lane * size.
// Issued in asm (M0 is compiler-reserved: written and restored inside the statement); the caller counts vmcnt itself.
__device__ __forceinline__ void at_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void at_dma4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---- attention dropout stream (all relpos attention kernels, forward and backward) ---------------------------------------------------
// The mask of score element (b, h, i, j) is a pure function of (call seed, b, h, i, j). A lane of these kernels owns, per block of 32
// keys jb .. jb+31, the 16 keys jb + 4*hh + 8*q + e (q, e = 0..3; hh = its half of the wave): eight 32-bit words, one per pair of
// consecutive keys. Word w of query row `row` = (b*H + h)*T + i:
//     y = S(row) + w * 0x9E3779B9 ;  y ^= y >> 15 ;  y = (y & 0xffffff) * 0x1b3c6d (low 32 bits) ;  y ^= y >> 16
// with S(row) = drop_hash(row, key) - the strong per-row hash of csrc/common.h - and w = ((jb/32)*2 + hh)*8 + 2*q + (e >> 1). The even key
// of the pair takes the low half of the word, the odd one the high half; a key is KEPT iff its half, read as a signed 16-bit number,
// is >= thr16 - 32768 (thr16 = round(p * 65536): the keep probability is (65536 - thr16) / 65536 exactly as in drop_keep_mask).
// One full-rate 24-bit multiply per word instead of drop_hash's two quarter-rate 32-bit ones: 3 VALU operations per element instead of
// ~9 (the hash was half of the forward's vector instructions). Statistics of the stream (keep rate, correlations between neighbours,
// rows, diagonals, words): tests/test_blocks_gpu.py::test_attention_dropout_stream_statistics; numpy twin: tests/helpers/attn_mask.py.
__device__ __forceinline__ unsigned attn_row_state(unsigned long long row, DropKey k) { return drop_hash(row, k); }
__device__ __forceinline__ unsigned attn_drop_word(unsigned row_state, unsigned w) {
    unsigned y = row_state + w * 0x9E3779B9u;
    y ^= y >> 15;
    y = __umul24(y, 0x1b3c6du);
    y ^= y >> 16;
    return y;
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
// 0xffff in each half of the word whose key is kept (two packed 16-bit operations: saturating subtract, arithmetic shift)
__device__ __forceinline__ unsigned attn_pair_mask(unsigned word, unsigned thr16) {
    const short c = (short)((int)thr16 - 32768 - 1);
    const s16x2 z = __builtin_elementwise_sub_sat((s16x2){c, c}, __builtin_bit_cast(s16x2, word));   // negative iff half > thr16 - 32768 - 1
    return __builtin_bit_cast(unsigned, z >> (s16x2){15, 15});
}
// the 8 pair masks of this lane for key block jblk (word k covers accumulator elements 2k, 2k+1)
__device__ __forceinline__ void attn_pair_masks(unsigned row_state, int jblk, int hh, unsigned thr16, unsigned (&m)[8]) {
    const unsigned w0 = (unsigned)(jblk * 2 + hh) * 8u;
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = attn_pair_mask(attn_drop_word(row_state, w0 + k), thr16);
}
// 16 keep bits (bit g = accumulator element g) out of the 8 pair masks
__device__ __forceinline__ unsigned attn_keep_bits(const unsigned (&m)[8]) {
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc |= m[k] & ((1u << (2 * k)) | (1u << (16 + 2 * k + 1)));
    return (acc | (acc >> 16)) & 0xffffu;
}
__device__ __forceinline__ unsigned attn_keep16(unsigned row_state, int jblk, int hh, unsigned thr16) {
    unsigned m[8];
    attn_pair_masks(row_state, jblk, hh, thr16, m);
    return attn_keep_bits(m);
}

// ---- helpers of the everything-in-LDS kernels (csrc/attention_short.hip, csrc/attention_fused.hip) ---------------------------------------

constexpr float AT2_THR_LOG2 = 6.f;     // the running maximum is raised when a new row maximum exceeds it by this many powers of two

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {     // own DMA pieces landed (all but the N youngest), then every wave's
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// x + (float) half of the packed fp16 pair, as fma(half, one, x) with `one` opaque to the optimiser: hipcc then selects v_fma_mix_f32 (the
// conversion rides in the operand). NOT inline asm: these read MFMA results, and hipcc pads the MFMA -> VALU wait states only for
// instructions it can see (guide 5.7 item 2) - the asm form of this file's first version read the accumulators early (NaN).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_h_lo(float x, unsigned pair, float one) { return __builtin_fmaf((float)__builtin_bit_cast(f16x2, pair)[0], one, x); }
__device__ __forceinline__ float add_h_hi(float x, unsigned pair, float one) { return __builtin_fmaf((float)__builtin_bit_cast(f16x2, pair)[1], one, x); }
__device__ __forceinline__ unsigned pk_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2)); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// one 8-row x 128-byte LDS-DMA piece of a [rows][64] bf16 tile through a buffer descriptor (base = the tile's row 0 / column 0 in global
// memory, 32-bit offsets): lane (prow = lane >> 3, pos = lane & 7) fetches 16 bytes of source row src_row + prow into LDS slot (prow, pos).
// voff_lane = prow * stride + (swizzled chunk << 4) is the lane's constant part (the caller keeps one per tile kind); a piece whose eight
// rows all exist costs one vector add on top of scalar arithmetic, a piece that straddles the first / last row clamps per lane.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_srd(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma_buf16(i32x4 srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_dst), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_piece(i32x4 srd, int stride_b, unsigned chunk_lane, int src_row, int lo, int hi, unsigned col_b, unsigned lds_dst, int lane) {
    // source row of this lane = clamp(src_row + prow, lo, hi) = src_row + med3(prow, lo - src_row, hi - src_row): three full-rate vector operations
    const int prow = lane >> 3;
    const int rel = min(max(prow, lo - src_row), hi - src_row);
    const unsigned voff = __umul24((unsigned)rel & 0xffffffu, (unsigned)stride_b) + chunk_lane + (unsigned)(src_row * stride_b);
    dma_buf16(srd, voff, col_b, lds_dst);
}
template <int OFF>
__device__ __forceinline__ void g_store_pair(unsigned addr, unsigned pair, bool dup) {      // fp16 pair -> rows OFF/64 and OFF/64 + 1 of a [32][32] fp16 tile
    asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(addr), "v"(pair), "n"(OFF), "n"(OFF + 64) : "memory");
    if (dup) asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(addr), "v"(pair), "n"(OFF + 4096), "n"(OFF + 4096 + 64) : "memory");
}

