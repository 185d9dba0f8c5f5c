// Shared device/host helpers for libtsasr_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tsasr_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ---- host-side error plumbing -------------------------------------------------------------
void tsasr_set_error(const char *fmt, ...);
#define TSASR_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            tsasr_set_error(__VA_ARGS__);     \
            return TSASR_E_INVALID;           \
        }                                     \
    } while (0)
#define TSASR_CHECK_LAUNCH(name)                                                     \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            tsasr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return TSASR_E_LAUNCH;                                                   \
        }                                                                            \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// csrc/reduce.hip: dst[c] (+)= sum_{p < nparts} src[p * pstride + c], c < width - queued while tsasr_reduce_defer(1) is in force
// (run by tsasr_reduce_flush in one launch), otherwise launched on `st`
bool tsasr_reduce_deferring();
void tsasr_reduce_submit(const float *src, float *dst, long long pstride, int nparts, int width, int accumulate, hipStream_t st);
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// storage-type generic loads/stores (T = float or bf16_t); all math is fp32
template <typename T> __device__ __forceinline__ float ld1(const T *p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T *p, float v) { *p = (T)v; }

// 8 consecutive elements -> fp32[8]
__device__ __forceinline__ void ld8(const float *p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p);
    const float4 b = *reinterpret_cast<const float4 *>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t *p, float (&o)[8]) {
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
    o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
    o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float *p, const float (&v)[8]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// two floats -> one register of two bf16 (round to nearest even) in ONE v_cvt_pk_bf16_f32. Element-wise `(bf16_t)x` casts come out as one
// conversion per element plus a v_perm per pair wherever the optimiser does not pair them itself (after selects, in long unrolled bodies)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); }
__device__ __forceinline__ bf16x8 bf16x8_of(unsigned a, unsigned b, unsigned c, unsigned d) {     // four converted pairs as an MFMA operand
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(bf16x8, (u32x4){a, b, c, d});
}
__device__ __forceinline__ void st8(bf16_t *p, const float (&v)[8]) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
}

// WRITE-THROUGH 16-byte store (sc1): the bytes leave the XCD's L2 for memory as they are produced instead of sitting there dirty until the
// end-of-kernel write-back, which the NEXT kernel of the chain waits for (guide "boundary" row: + dirty bytes / 6 TB/s per dependent
// launch; the reader is usually on another XCD and reads from beyond L2 anyway). Inline asm: hipcc does not count it - the in-order vmcnt
// only makes its own later waits wait longer, never shorter.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_wt(void *p, u32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st8_wt(bf16_t *p, const float (&v)[8]) {
    st16_wt(p, (u32x4_t){pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])});
}
// st8_g / st16_g: 16-byte stores of a LARGE kernel output that the next kernel streams once (the 32.8 MB FFN activation / its gradient:
// csrc/gemm_big.hip) - write-through unless the library is built with -DTSASR_NO_WT (`make nowt`, the A/B build of profiles/r04_notes.md:
// 10.79 -> 10.70 ms per step). NOT for the 4 MB row-kernel / projection outputs: written through they cost the step 0.07 ms (their
// readers find them in L2 when they stay).
#ifdef TSASR_NO_WT
__device__ __forceinline__ void st8_g(bf16_t *p, const float (&v)[8]) { st8(p, v); }
__device__ __forceinline__ void st16_g(void *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }
#else
__device__ __forceinline__ void st8_g(bf16_t *p, const float (&v)[8]) { st8_wt(p, v); }
__device__ __forceinline__ void st16_g(void *p, uint4 v) { st16_wt(p, (u32x4_t){v.x, v.y, v.z, v.w}); }
#endif

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// ---- counter-based dropout bits ------------------------------------------------------------
// The mask of an element is a pure function of (64-bit call seed, element index): never stored, regenerated by every backward
// kernel. One 32-bit integer hash (two v_mul_lo_u32 + shifts/xors; the 64-bit splitmix finaliser used before cost more than
// the GEMM it was fused into) yields two 16-bit uniforms; an element is kept iff its 16 bits >= round(p * 65536), and kept
// values are scaled by 65536 / (65536 - threshold) so that the expectation is exact for the quantised probability.
struct DropKey { unsigned k0, k1; };
__device__ __forceinline__ DropKey drop_key(unsigned long long seed) {   // once per thread, not per element
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return DropKey{(unsigned)z, (unsigned)(z >> 32)};
}
__device__ __forceinline__ unsigned drop_hash(unsigned long long ctr, DropKey k) {
    unsigned x = (unsigned)ctr + k.k0;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x += k.k1 ^ (unsigned)(ctr >> 32);
    x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_thr16(float p) { return p > 0.f ? min(65535u, (unsigned)(p * 65536.f + 0.5f)) : 0u; }
__device__ __forceinline__ float drop_scale16(unsigned thr16) { return 65536.f / (float)(65536u - thr16); }
// keep-bits of N consecutive elements starting at the EVEN index idx (bit j = element idx + j kept); N even
template <int N>
__device__ __forceinline__ unsigned drop_keep_mask(unsigned long long idx, DropKey k, unsigned thr16) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        const unsigned h = drop_hash((idx + j) >> 1, k);
        m |= ((h & 0xffffu) >= thr16 ? 1u : 0u) << j;
        m |= ((h >> 16) >= thr16 ? 1u : 0u) << (j + 1);
    }
    return m;
}
// keep-bits of the FOUR consecutive elements idx0 .. idx0 + 3 (bit e = element idx0 + e), any idx0: the same bits drop_keep1 gives each
// of them, from two hashes (three when idx0 is odd) instead of four - a hash is two quarter-rate integer multiplies plus ~8 VALU
// operations, and the attention kernels spend as many cycles on these as on their MFMAs
__device__ __forceinline__ unsigned drop_keep4(unsigned long long idx0, DropKey k, unsigned thr16) {
    const unsigned long long c0 = idx0 >> 1;
    const unsigned h0 = drop_hash(c0, k), h1 = drop_hash(c0 + 1, k);
    const unsigned k0 = (h0 & 0xffffu) >= thr16, k1 = (h0 >> 16) >= thr16, k2 = (h1 & 0xffffu) >= thr16, k3 = (h1 >> 16) >= thr16;
    if (!((unsigned)idx0 & 1u)) return k0 | (k1 << 1) | (k2 << 2) | (k3 << 3);
    const unsigned h2 = drop_hash(c0 + 2, k);
    const unsigned k4 = (h2 & 0xffffu) >= thr16;
    return k1 | (k2 << 1) | (k3 << 2) | (k4 << 3);
}
// single element (any index): the same bit drop_keep_mask assigns to it
__device__ __forceinline__ bool drop_keep1(unsigned long long idx, DropKey k, unsigned thr16) {
    const unsigned h = drop_hash(idx >> 1, k);
    return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}

// wave-wide reductions over all 64 lanes (result in every lane). DPP inside each 16-lane row (quad swaps, then the two
// mirrors: after the quad steps the partial values are quad-uniform, so lane i + lane 15-i covers the row), then the four
// row values are combined through v_readlane. ~10 short-latency VALU ops instead of six dependent ds_bpermute round trips
// (__shfl_xor goes through the LDS crossbar: ~1.4k cycles per pair of sums in the one-wave-per-row LayerNorm kernels).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// value of the lane 32 positions away (the other half of the wave): one v_permlane32_swap_b32 (gfx950) instead of a ds_bpermute
// round trip through the LDS crossbar
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: lanes 32..63 <- lanes 0..31 ; r[1]: lanes 0..31 <- lanes 32..63
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
// FOUR wave-wide sums at once (every lane gets all four): the partial sums are folded into one register as soon as two lanes hold the same
// value (after the first quad step lane parity picks which pair of values a lane carries on, after the second bit 1 of the lane), the rows
// are joined by v_permlane16_swap / v_permlane32_swap (gfx950: swap(x, x) returns both rows' / halves' values, their sum is the same in both),
// and four quad broadcasts hand the totals back: 19 vector operations instead of 4 x 11; fixed order.
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += dpp_mov<0xB1>(v[q]);                   // quad_perm [1,0,3,2]
    float a = (lane & 1) ? v[1] : v[0], b = (lane & 1) ? v[3] : v[2];
    a += dpp_mov<0x4E>(a);                                                     // quad_perm [2,3,0,1]
    b += dpp_mov<0x4E>(b);
    float c = (lane & 2) ? b : a;                                              // lane & 3 = index of the value this lane carries
    c += dpp_mov<0x124>(c);                                                    // row_ror:4
    c += dpp_mov<0x128>(c);                                                    // row_ror:8
    // (inline asm: through __builtin_amdgcn_permlane16_swap / 32_swap hipcc (ROCm 7.2) folds r[0] + r[1] into 2 * r[0] - it takes the two
    // results of the swap for equal. `s_nop 1`: the wait states hipcc itself puts between a VALU write and a lane swap of the same register)
    {
        float c2 = c;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(c2));
        c += c2;
        c2 = c;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(c2));
        c += c2;
    }
    v[0] = dpp_mov<0x00>(c); v[1] = dpp_mov<0x55>(c); v[2] = dpp_mov<0xAA>(c); v[3] = dpp_mov<0xFF>(c);
}
// sum over the 32 lanes of this lane's half of the wave (two independent rows per wave: lanes 0-31 / 32-63)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    const float lo = lane_bcast(v, 0) + lane_bcast(v, 16), hi = lane_bcast(v, 32) + lane_bcast(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
