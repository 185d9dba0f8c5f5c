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
__device__ __forceinline__ void st8(bf16_t *p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8 *>(p) = o;
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// wave-wide reductions over all 64 lanes (butterfly; result in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
