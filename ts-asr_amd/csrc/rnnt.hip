// RNN-T joint + head + loss for gfx950 (MI355X).
//
// Replaces (reference paths; SB = vendor/speechbrain/speechbrain):
//   SB/nnet/transducer/transducer_joint.py:73-95 + SB/nnet/linear.py:64-78   (joint "sum" + LeakyReLU + head)
//   SB/nnet/losses.py:29-87 -> torchaudio.functional.rnnt_loss                (default loss path)
//   SB/nnet/loss/transducer_loss.py:31-236                                    (the same lattice as Numba spin-lock kernels)
//
// Design (MI355X-first, not a translation of the Numba kernels):
//   * joint_fwd: the [B,T,U1,J] joint tensor (2.48 GB fp32 in the reference) is never built. A workgroup owns
//     (b, 32 lattice columns u); the dec tile and the head matrix live in LDS as bf16, each wave walks time
//     frames, forms h = lrelu(enc[t]+dec[u]) in registers and feeds v_mfma_f32_32x32x16_bf16 with
//     A = W (rows = vocabulary), B = h^T (cols = u) so that every lane ends up with 16 vocabulary entries
//     of ONE lattice cell -> 16-byte row stores into logits rows padded to `ldl` floats.
//   * loss: (1) rnnt_lp: fused log-softmax, 8 lanes per lattice cell, keeps only lse / lp(blank) / lp(label);
//           (2) rnnt_alphabeta: ONE WAVE per (utterance, direction): lane l owns K consecutive columns, lanes
//               are skewed by one time step, the neighbour's boundary value moves with a DPP wave shift ->
//               no LDS, no barrier, no atomics (the reference serialises columns with an int32 spin-lock);
//           (3) rnnt_grad: gradient w.r.t. logits (softmax * occupancy - transitions), written once.
//   * joint_bwd: two deterministic kernels that both recompute dh = lrelu'(enc+dec) * (dlogits . W) on MFMA:
//       X: workgroup (b, u-tile) walks t  -> ddec, head dW/dbias slabs (reduced by a tiny second kernel);
//       Y: workgroup (b, 8 frames) walks u-tiles -> denc.  No float atomics anywhere => bitwise reproducible.
#include <type_traits>

#include <stdlib.h>

#include <algorithm>

#include "common.h"

#define NEG_INF (-INFINITY)

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// ============================================================================================
// joint forward
// ============================================================================================
// LDS row stride (in bf16) for a J-wide row: +8 elements (16 B) so that 16 consecutive rows start in
// 16 different 16-byte slots of the 256-byte bank row (ds_read_b128 conflict-free; guide G4).
__host__ __device__ static inline int lds_stride(int J) { return J + 8; }

template <typename T>
__global__ __launch_bounds__(256) void joint_fwd_kernel(const T *__restrict__ enc, const T *__restrict__ dec,
                                                        const float *__restrict__ W, const float *__restrict__ bias,
                                                        float *__restrict__ logits, int Tn, int U1, int J, int V,
                                                        int ldl, float slope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = lds_stride(J);
    bf16_t *w_lds = reinterpret_cast<bf16_t *>(smem);     // [32][S] bf16 (MFMA operand)
    T *d_lds = reinterpret_cast<T *>(w_lds + 32 * S);     // [32][S] in the activation storage type (no extra rounding)
    const int b = blockIdx.z, u0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // stage W (fp32 master -> bf16) and the dec tile
    for (int i = tid; i < 32 * (J / 8); i += 256) {
        const int row = i / (J / 8), c = (i % (J / 8)) * 8;
        float w8[8];
        if (row < V) ld8(W + (size_t)row * J + c, w8);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = 0.f;
        }
        st8(w_lds + row * S + c, w8);
        float d8[8];
        if (u0 + row < U1) ld8(dec + ((size_t)b * U1 + u0 + row) * J + c, d8);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) d8[j] = 0.f;
        }
        st8(d_lds + row * S + c, d8);
    }
    __syncthreads();

    // bias for this lane's 16 vocabulary rows
    float bv[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int v = (g & 3) + 8 * (g >> 2) + 4 * h;
        bv[g] = v < V ? bias[v] : 0.f;
    }
    const int tchunk = cdiv_dev(Tn, (int)gridDim.y);
    const int t_begin = blockIdx.y * tchunk, t_end = min(Tn, t_begin + tchunk);
    const bf16_t *wrow = w_lds + r * S + 8 * h;
    const T *drow = d_lds + r * S + 8 * h;
    const int nks = J / 16;
    // The enc row of a frame is the same for all 64 lanes: it is fetched ONCE per wave (coalesced 16-byte pieces, one frame
    // ahead, held in registers across the MFMA loop) into a per-wave LDS row and read from there as broadcasts, instead of
    // every lane re-requesting it from global memory in every k-step (40 exposed round trips per frame).
    constexpr int VE = 16 / sizeof(T);
    T *e_w = d_lds + 32 * S + wave * J;                   // [4 waves][J]
    uint4 stg[3];                                         // J <= 3 * 64 * VE (checked on the host)
    auto request = [&](int t) {
        const T *src = enc + ((size_t)b * Tn + min(t, Tn - 1)) * J;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int c = (q * 64 + lane) * VE;
            stg[q] = *reinterpret_cast<const uint4 *>(src + min(c, J - VE));
        }
    };
    request(t_begin + wave);
    for (int t = t_begin + wave; t < t_end; t += 4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int c = (q * 64 + lane) * VE;
            if (c < J) *reinterpret_cast<uint4 *>(e_w + c) = stg[q];
        }
        request(t + 4);                                   // next frame of this wave: in flight during the k-loop below
        __builtin_amdgcn_wave_barrier();
        const T *erow = e_w + 8 * h;
        f32x16 acc = {0};
#pragma unroll 4
        for (int s = 0; s < nks; ++s) {
            float e8[8], d8[8];
            ld8(erow + 16 * s, e8);
            ld8(drow + 16 * s, d8);
            bf16x8 hb;
#pragma unroll
            for (int j = 0; j < 8; ++j) hb[j] = (bf16_t)lrelu(e8[j] + d8[j], slope);
            const bf16x8 wa = *reinterpret_cast<const bf16x8 *>(wrow + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb, acc, 0, 0, 0);
        }
        if (u0 + r < U1) {
            float *orow = logits + (((size_t)b * Tn + t) * U1 + u0 + r) * ldl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // vocabulary rows 8q+4h .. 8q+4h+3 -> one 16-byte store
                const int v0 = 8 * q + 4 * h;
                if (v0 < ldl) {
                    float4 o;
                    o.x = (v0 + 0 < V) ? acc[4 * q + 0] + bv[4 * q + 0] : 0.f;
                    o.y = (v0 + 1 < V) ? acc[4 * q + 1] + bv[4 * q + 1] : 0.f;
                    o.z = (v0 + 2 < V) ? acc[4 * q + 2] + bv[4 * q + 2] : 0.f;
                    o.w = (v0 + 3 < V) ? acc[4 * q + 3] + bv[4 * q + 3] : 0.f;
                    *reinterpret_cast<float4 *>(orow + v0) = o;
                }
            }
        }
    }
}

// The same forward with the head matrix in REGISTERS (J = 16 * NKS known at compile time, bf16 activations): a lane's 16-byte W pieces for
// all NKS k-steps are loaded once (4 * NKS VGPRs), which (a) removes one of the three LDS reads of every k-step and (b) halves the
// workgroup's LDS (dec tile + one enc row per wave: 47 KB at J = 640), so two workgroups share a CU and a SIMD holds two waves - the
// one-wave form left the LDS round trips and the MFMA issue of a frame uncovered (the kernel is bound by the VALU work of
// h = lrelu(enc + dec), 5.5 operations per lattice cell and joint dimension).
// S01: 0 <= slope <= 1 (checked by the launcher; LeakyReLU's 0.01): lrelu(x) = max(x, slope * x), two operations instead of three - the same
// values (x > 0: x >= slope * x; x <= 0: slope * x >= x).
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
// 8 bf16 (one 16-byte word) -> 8 fp16 (exact for |x| < 65504 with <= 11 significant bits: every bf16 value in that range)
__device__ __forceinline__ uint4 bf16x8_to_f16x8(uint4 v) {
    auto cv = [](unsigned w) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}, half2_t)); };
    return make_uint4(cv(v.x), cv(v.y), cv(v.z), cv(v.w));
}
template <int NKS, bool S01>
__global__ __launch_bounds__(256, 2) void joint_fwd_regw_kernel(const bf16_t *__restrict__ enc, const bf16_t *__restrict__ dec,
                                                                const float *__restrict__ W, const float *__restrict__ bias,
                                                                float *__restrict__ logits, int Tn, int U1, int V, int ldl, float slope) {
    // Round 5: the hidden activation h = lrelu(enc + dec) is formed in PACKED fp16 (v_pk_add_f16 / v_pk_mul_f16 / v_pk_max_f16: 1.5
    // vector operations per lattice cell and joint dimension instead of 5.5 - the bf16 tiles had to be unpacked to fp32, added, scaled,
    // maxed and packed again) and the head GEMM runs on v_mfma_f32_32x32x16_f16. enc / dec / W are converted to fp16 ONCE when they are staged
    // (bf16 -> fp16 is exact below 65504; the projections' outputs are O(1) - a value beyond fp16's range would saturate to inf, bf16
    // kept it); x = e + d and slope * x are rounded to 11 significant bits (the bf16 form rounded h to 8).
    constexpr int J = 16 * NKS, S = J + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *d_lds = reinterpret_cast<_Float16 *>(smem);     // [32][S]
    const int b = blockIdx.z, u0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // the head matrix goes through the (still unused) dec-tile area once: fp32 -> fp16 rows in LDS, from there this lane's NKS pieces
    // into registers (a direct per-lane fetch of 8 * NKS fp32 values had all of them in flight at once and spilled)
    for (int i = tid; i < 32 * (J / 8); i += 256) {
        const int row = i / (J / 8), c = (i % (J / 8)) * 8;
        float w8[8];
        ld8(W + (size_t)min(row, V - 1) * J + c, w8);
        if (row >= V) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = 0.f;
        }
        half8_t wh;
#pragma unroll
        for (int j = 0; j < 8; ++j) wh[j] = (_Float16)w8[j];
        *reinterpret_cast<half8_t *>(d_lds + row * S + c) = wh;
    }
    __syncthreads();
    half8_t wreg[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) wreg[s] = *reinterpret_cast<const half8_t *>(d_lds + r * S + 8 * h + 16 * s);
    __syncthreads();
    for (int i = tid; i < 32 * (J / 8); i += 256) {
        const int row = i / (J / 8), c = (i % (J / 8)) * 8;
        uint4 v = *reinterpret_cast<const uint4 *>(dec + ((size_t)b * U1 + min(u0 + row, U1 - 1)) * J + c);
        if (u0 + row >= U1) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(d_lds + row * S + c) = bf16x8_to_f16x8(v);
    }
    float *b_lds = reinterpret_cast<float *>(d_lds + 32 * S + 4 * J);     // [32] head bias (0 beyond V): read per frame, not held in 16 registers
    if (tid < 32) b_lds[tid] = bias[min(tid, V - 1)] * (tid < V ? 1.f : 0.f);
    __syncthreads();
    const int tchunk = cdiv_dev(Tn, (int)gridDim.y);
    const int t_begin = blockIdx.y * tchunk, t_end = min(Tn, t_begin + tchunk);
    const _Float16 *drow = d_lds + r * S + 8 * h;
    _Float16 *e_w = d_lds + 32 * S + wave * J;            // [4 waves][J]: the frame's enc row, fetched once per wave, read as broadcasts
    constexpr int NQ = (J + 511) / 512;                   // 16-byte pieces per lane
    uint4 stg[NQ];
    auto request = [&](int t) {
        const bf16_t *src = enc + ((size_t)b * Tn + min(t, Tn - 1)) * J;
#pragma unroll
        for (int q = 0; q < NQ; ++q) stg[q] = *reinterpret_cast<const uint4 *>(src + min((q * 64 + lane) * 8, J - 8));
    };
    const _Float16 sl16 = (_Float16)slope;
    const half8_t slope8 = {sl16, sl16, sl16, sl16, sl16, sl16, sl16, sl16};
    request(t_begin + wave);
    for (int t = t_begin + wave; t < t_end; t += 4) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 8;
            if (c < J) *reinterpret_cast<uint4 *>(e_w + c) = bf16x8_to_f16x8(stg[q]);
        }
        request(t + 4);
        __builtin_amdgcn_wave_barrier();
        const _Float16 *erow = e_w + 8 * h;
        f32x16 acc = {0};
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const half8_t e8 = *reinterpret_cast<const half8_t *>(erow + 16 * s), d8 = *reinterpret_cast<const half8_t *>(drow + 16 * s);
            const half8_t x = e8 + d8;
            half8_t hb;
            if (S01) hb = __builtin_elementwise_max(x, x * slope8);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) hb[j] = x[j] > (_Float16)0 ? x[j] : x[j] * sl16;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], hb, acc, 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting all 2 * NKS LDS reads of a frame (it spilled 149 VGPRs)
        }
        __builtin_amdgcn_wave_barrier();                  // this frame's reads of e_w are issued before the next frame's row lands in it
        if (u0 + r < U1) {
            float *orow = logits + (((size_t)b * Tn + t) * U1 + u0 + r) * ldl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int v0 = 8 * q + 4 * h;
                if (v0 < ldl) {
                    const float4 bq = *reinterpret_cast<const float4 *>(b_lds + v0);
                    float4 o;
                    o.x = (v0 + 0 < V) ? acc[4 * q + 0] + bq.x : 0.f;
                    o.y = (v0 + 1 < V) ? acc[4 * q + 1] + bq.y : 0.f;
                    o.z = (v0 + 2 < V) ? acc[4 * q + 2] + bq.z : 0.f;
                    o.w = (v0 + 3 < V) ? acc[4 * q + 3] + bq.w : 0.f;
                    *reinterpret_cast<float4 *>(orow + v0) = o;
                }
            }
        }
    }
}

// ============================================================================================
// joint backward: shared "masked dh tile" machinery
// ============================================================================================
#ifndef KB
#define KB 3  // 32-wide k-blocks per wave (measured on the step: 2 -> 0.87 ms, 3 -> 0.67, 4 -> 0.72, 5 -> 1.14)
#endif

struct BwdFrags {
    bf16x8 wf[KB][2];  // B operand of D = dlogits . W : W[v = 16s+8h+j][k_r + 32 kb]
    int kb[KB];        // global k-block index or -1
};

__device__ __forceinline__ void load_w_frags(BwdFrags &f, const float *__restrict__ W, int J, int V, int wslot,
                                             int nslots, int r, int h) {
    const int nkb = J / 32;
    // every element is ALWAYS requested (vocabulary row / k-block clamped into the matrix) and masked afterwards: a guarded load sits in a
    // basic block of its own and is waited for on the spot - the 48 of them were 48 serialized round trips at the head of every workgroup
    float wv[KB][2][8];
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        const int kb = wslot + i * nslots, kbc = min(kb, nkb - 1);
        f.kb[i] = kb < nkb ? kb : -1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[i][s][j] = W[(size_t)min(16 * s + 8 * h + j, V - 1) * J + kbc * 32 + r];
    }
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                f.wf[i][s][j] = (bf16_t)((f.kb[i] >= 0 && 16 * s + 8 * h + j < V) ? wv[i][s][j] : 0.f);
}

// the same fragments in fp16, straight from the fp32 head matrix (the forward's operand rounding: joint_fwd_regw_kernel)
typedef _Float16 half8_w __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void load_w_frags16(half8_w (&wf)[KB][2], const BwdFrags &f, const float *__restrict__ W, int J, int V, int wslot, int nslots, int r,
                                               int h) {
    const int nkb = J / 32;
    float wv[KB][2][8];
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        const int kbc = min(wslot + i * nslots, nkb - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[i][s][j] = W[(size_t)min(16 * s + 8 * h + j, V - 1) * J + kbc * 32 + r];
    }
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[i][s][j] = (_Float16)((f.kb[i] >= 0 && 16 * s + 8 * h + j < V) ? wv[i][s][j] : 0.f);
}

// A operand: dlogits[u_r][16s+8h+j] (fp32 -> bf16); rows beyond U1 read as zero
__device__ __forceinline__ void load_a_frags(const float *__restrict__ dl_row, bool valid, int ldl, int h,
                                             bf16x8 (&a)[2], float (&af)[2][8]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int v0 = 16 * s + 8 * h;
        if (valid && v0 + 8 <= ldl) ld8(dl_row + v0, af[s]);
        else if (valid && v0 + 4 <= ldl) {
            const float4 x = *reinterpret_cast<const float4 *>(dl_row + v0);
            af[s][0] = x.x; af[s][1] = x.y; af[s][2] = x.z; af[s][3] = x.w;
            af[s][4] = af[s][5] = af[s][6] = af[s][7] = 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) af[s][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[s][j] = (bf16_t)af[s][j];
    }
}

// Unconditional forms for the software-pipelined loops (ldl == 32: a dlogits row is one 128-byte line). A conditional load sits
// in its own basic block and is waited for on the spot - s_memtime stamps showed 90% of both backward kernels in exposed load
// latency - so addresses are clamped, the load is always issued, and invalid rows are zeroed when the fragment is built.
__device__ __forceinline__ void load_a_raw32(const float *__restrict__ dl_row, int h, float (&af)[2][8]) {
    ld8(dl_row + 8 * h, af[0]);
    ld8(dl_row + 16 + 8 * h, af[1]);
}
__device__ __forceinline__ void cvt_a(float (&af)[2][8], bool valid, bf16x8 (&a)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            af[s][j] = valid ? af[s][j] : 0.f;
            a[s][j] = (bf16_t)af[s][j];
        }
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_j;

// u index (within the 32-row tile) of accumulator register g for lane half h (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// ---- X: ddec + head-weight slabs ------------------------------------------------------------
#ifndef JBX_OCC
#define JBX_OCC 1   // workgroups of joint_bwd_x_kernel a CU is meant to hold (lab builds: -DKB=2 -DJBX_OCC=2)
#endif
template <typename T>
__global__ __launch_bounds__(256, JBX_OCC) void joint_bwd_x_kernel(
    const float *__restrict__ dlogits, const T *__restrict__ enc, const T *__restrict__ dec,
    const float *__restrict__ W, T *__restrict__ ddec, float *__restrict__ slab_w /*[B*nut][32][J]*/,
    float *__restrict__ slab_b /*[B*nut][32]*/, const int32_t *__restrict__ tlen, const int32_t *__restrict__ ulen,
    int Tn, int U1, int J, int V, int ldl, float slope, float *__restrict__ denc_part /*[nut][B][T][J] or NULL*/,
    int TS, float *__restrict__ ddec_part /*[TS][B][U1][J], TS > 1*/) {
    __shared__ __attribute__((aligned(16))) bf16_t a_lds[4][32 * 32];  // per-wave dlogits tile [u][v]
    // blockIdx.z = b * TS + ts: the frames are cut into TS ranges (one workgroup each) when B * tiles alone leave CUs with a single
    // workgroup (4 waves at 180 VGPRs: nothing to switch to while a frame's loads are in flight); the ddec / dW / dbias partial sums
    // of the ranges are added in fixed order afterwards
    const int nb = gridDim.z / TS, b = blockIdx.z / TS, ts = blockIdx.z - b * TS, ut = blockIdx.x, u0 = ut * 32, nut = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nslots = gridDim.y * 4, wslot = blockIdx.y * 4 + wave;
    const int Tb = tlen ? min(max(tlen[b], 1), Tn) : Tn;
    const int Ub = ulen ? min(max(ulen[b], 0), U1 - 1) : U1 - 1;
    const int t_chunk = (Tn + TS - 1) / TS, t_lo = ts * t_chunk, t_hi = min(Tn, t_lo + t_chunk);
    const int t_end = (u0 <= Ub) ? min(Tb, t_hi) : t_lo;  // tile entirely outside the lattice -> dlogits are zero

    // F16 (bf16 training path, round 5): the vector work per lattice cell and joint dimension in PACKED fp16, as the forward does it -
    //   x = e + d, h = max(x, slope x), lrelu'(x) = bit-select(sign mask of x, slope, 1), dh = cvt(D) * lrelu'  : 6 packed operations + one
    //   conversion per PAIR of cells instead of ~7 fp32 operations per cell -
    // with every MFMA on fp16 operands (dlogits and W in fp16: 11 significant bits instead of bf16's 8) and the two sums of dh off the vector
    // unit: over the frames (ddec) as an MFMA against an identity operand with dh, in the registers it was formed in, as B operand; over the
    // tile's rows (denc) as 7 packed adds. (fp32 tensors take the fp32 form below.)
    constexpr bool F16 = !std::is_same<T, float>::value;
    BwdFrags f;
    load_w_frags(f, W, J, V, wslot, nslots, r, h);
    float dv[KB][16], dacc[KB][16];
    half2_t dvp[KB][8];
    half8_t wf16[KB][2], ident16[2];
    f32x16 wacc[KB], dsum[KB];
    int kbc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) kbc[i] = max(f.kb[i], 0);
    {
        T raw[KB][16];
#pragma unroll
        for (int i = 0; i < KB; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) raw[i][g] = dec[((size_t)b * U1 + min(u0 + acc_row(g, h), U1 - 1)) * J + kbc[i] * 32 + r];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            wacc[i] = (f32x16){0};
            dsum[i] = (f32x16){0};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                dv[i][g] = (f.kb[i] >= 0 && u0 + acc_row(g, h) < U1) ? (float)raw[i][g] : 0.f;
                dacc[i][g] = 0.f;
            }
#pragma unroll
            for (int pq = 0; pq < 8; ++pq) dvp[i][pq] = (half2_t){(_Float16)dv[i][2 * pq], (_Float16)dv[i][2 * pq + 1]};
        }
        if constexpr (F16) load_w_frags16(wf16, f, W, J, V, wslot, nslots, r, h);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) ident16[s2][j] = (_Float16)((r == 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)) ? 1.f : 0.f);
    }
    float bsum[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[s][j] = 0.f;

    bf16_t *my_lds = a_lds[wave];
    const bool row_ok = (u0 + r) < U1;
    // one frame: a = this lane's dlogits fragments (bf16), af = the same values in fp32 (bias-gradient sums), ev = enc[b, t, k] per k-block
    const half2_t slope2 = {(_Float16)slope, (_Float16)slope};
    const unsigned slope2u = __builtin_bit_cast(unsigned, slope2), one2u = 0x3C003C00u;
    auto frame = [&](int t, const bf16x8 (&a)[2], const float (&af)[2][8], const float (&ev)[KB]) {
        half8_t a16[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[s][j] += af[s][j];
            if constexpr (F16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a16[s][j] = (_Float16)af[s][j];      // (af is already zero for rows beyond U1: cvt_a)
                *reinterpret_cast<half8_t *>(my_lds + r * 32 + 16 * s + 8 * h) = a16[s];
            } else {
                *reinterpret_cast<bf16x8 *>(my_lds + r * 32 + 16 * s + 8 * h) = a[s];
            }
        }
        __builtin_amdgcn_wave_barrier();  // same-wave LDS write -> read (DS ops retire in order)
        // A' operand of dW = dlogits^T . H : rows = v_r, inner index i <-> u = 16s + 8(j>>2) + 4h + (j&3)
        // (two transposing LDS reads per fragment: the tile is [u][v] with 64-byte rows, lane (v = r, h) needs u = 16s + 8(j>>2) + 4h + (j&3);
        //  sixteen 2-byte reads, waited for in pairs, were eight serialized LDS round trips per frame)
        bf16x8 at[2];
        {
            const int q4 = (lane & 15) >> 2, p4 = lane & 3, vhalf = (lane >> 4) & 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16_t *a0 = my_lds + (16 * s + 4 * h + q4) * 32 + 16 * vhalf + 4 * p4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_j *)(a0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_j *)(a0 + 8 * 32));
                at[s][0] = lo[0]; at[s][1] = lo[1]; at[s][2] = lo[2]; at[s][3] = lo[3];
                at[s][4] = hi[0]; at[s][5] = hi[1]; at[s][6] = hi[2]; at[s][7] = hi[3];
            }
        }
        if constexpr (F16) {
            half8_t at16[2];
            at16[0] = __builtin_bit_cast(half8_t, at[0]);
            at16[1] = __builtin_bit_cast(half8_t, at[1]);
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                if (f.kb[i] < 0) continue;   // wave-uniform
                const _Float16 e16 = (_Float16)ev[i];
                const half2_t e2 = {e16, e16};
                f32x16 D = {0};
                D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[0], wf16[i][0], D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[1], wf16[i][1], D, 0, 0, 0);
                typedef unsigned u32x4_j __attribute__((ext_vector_type(4)));
                typedef short s16x2_j __attribute__((ext_vector_type(2)));
                u32x4_j hbv[2], dbv[2];      // the B operands are written element by element where they will be read (no gathering moves)
                half2_t dhp[8];
#pragma unroll
                for (int pq = 0; pq < 8; ++pq) {
                    const half2_t x = dvp[i][pq] + e2, xs = x * slope2;
                    // 0xffff in the halves with x <= 0 (torch: lrelu'(x) = x > 0 ? 1 : slope - and with bf16 operands e = -d happens once in ~10^4 cells):
                    // as int16, fp16 bits are <= 0 exactly for -x and +0, so (bits - 1, saturating) is negative there and nowhere else
                    const unsigned neg = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_j, x), (s16x2_j){1, 1}) >> (s16x2_j){15, 15});
                    // lrelu(x) and lrelu'(x) as bit-selects on that mask; v_bfi_b32 by hand: left to itself hipcc makes two 16-bit compare + select pairs
                    // and a v_perm (or not / and / and-or) out of the first one
                    unsigned hsel, fct;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hsel) : "v"(neg), "v"(__builtin_bit_cast(unsigned, xs)), "v"(__builtin_bit_cast(unsigned, x)));
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(fct) : "v"(neg), "v"(slope2u), "v"(one2u));
                    hbv[pq >> 2][pq & 3] = hsel;
                    const half2_t dp = __builtin_convertvector((f32x2){D[2 * pq], D[2 * pq + 1]}, half2_t);
                    dhp[pq] = dp * __builtin_bit_cast(half2_t, fct);
                    dbv[pq >> 2][pq & 3] = __builtin_bit_cast(unsigned, dhp[pq]);
                }
                half8_t hb16[2], db16[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    hb16[s2] = __builtin_bit_cast(half8_t, hbv[s2]);
                    db16[s2] = __builtin_bit_cast(half8_t, dbv[s2]);
                }
                dsum[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ident16[0], db16[0], dsum[i], 0, 0, 0);
                dsum[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ident16[1], db16[1], dsum[i], 0, 0, 0);
                if (denc_part) {      // this tile's share of denc[b, t, k]: the 16 rows of the lane as packed adds, the other half-wave's 16 by a lane swap
                    const half2_t tsum = ((dhp[0] + dhp[1]) + (dhp[2] + dhp[3])) + ((dhp[4] + dhp[5]) + (dhp[6] + dhp[7]));
                    float esum = (float)tsum[0] + (float)tsum[1];
                    esum += other_half(esum);
                    if (h == 0) denc_part[(((size_t)ut * nb + b) * Tn + t) * J + f.kb[i] * 32 + r] = esum;
                }
                wacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at16[0], hb16[0], wacc[i], 0, 0, 0);
                wacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at16[1], hb16[1], wacc[i], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if (f.kb[i] < 0) continue;   // wave-uniform
            const float e = ev[i];
            f32x16 D = {0};
            D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f.wf[i][0], D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], f.wf[i][1], D, 0, 0, 0);
            bf16x8 hb[2];
            float esum = 0.f;   // sum of dh over this tile's 16 u rows of the lane: the tile's share of denc[b, t, k]
            float hv[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float x = e + dv[i][g];
                const float fct = x > 0.f ? 1.f : slope;      // one select serves the activation and its derivative (x * 1 and D * 1 are exact)
                const float dh = D[g] * fct;
                dacc[i][g] += dh;
                esum += dh;
                hv[g] = x * fct;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                hb[s2] = bf16x8_of(pk_bf16(hv[8 * s2], hv[8 * s2 + 1]), pk_bf16(hv[8 * s2 + 2], hv[8 * s2 + 3]), pk_bf16(hv[8 * s2 + 4], hv[8 * s2 + 5]),
                                   pk_bf16(hv[8 * s2 + 6], hv[8 * s2 + 7]));
            if (denc_part) {    // dh is in registers here anyway: the separate denc pass recomputed every one of these tiles
                esum += other_half(esum);
                if (h == 0) denc_part[(((size_t)ut * nb + b) * Tn + t) * J + f.kb[i] * 32 + r] = esum;
            }
            wacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], hb[0], wacc[i], 0, 0, 0);
            wacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], hb[1], wacc[i], 0, 0, 0);
        }
    };
    if (ldl == 32) {
        // software pipeline over the frames (a dlogits row is one 128-byte line): the dlogits fragments and enc values of frame t + NP are
        // requested when frame t's have been converted, NP frames ahead - a frame is ~1 us of issue for the one wave a SIMD holds here,
        // a round trip to HBM about twice that (one frame ahead left half of every frame waiting: 0.56 ms per launch at configs[1])
        constexpr int NP = 3;     // measured at configs[1] (ms per step / joint + loss): 2 -> 12.34 / 0.93, 3 -> 12.24 / 0.87, 4 -> 12.27 / 0.89, 6 -> 12.32 / 0.93
        const float *dl0 = dlogits + (((size_t)b * Tn) * U1 + min(u0 + r, U1 - 1)) * 32;
        float sa[NP][2][8];
        T se[NP][KB];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int tt = min(t_lo + s, Tn - 1);
            load_a_raw32(dl0 + (size_t)tt * U1 * 32, h, sa[s]);
#pragma unroll
            for (int i = 0; i < KB; ++i) se[s][i] = enc[((size_t)b * Tn + tt) * J + kbc[i] * 32 + r];
        }
        for (int t0 = t_lo; t0 < t_end; t0 += NP) {
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const int t = t0 + s;
                if (t < t_end) {          // wave-uniform
                    bf16x8 a[2];
                    float af[2][8], ev[KB];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) af[s2][j] = sa[s][s2][j];
#pragma unroll
                    for (int i = 0; i < KB; ++i) ev[i] = (float)se[s][i];
                    const int tn = min(t + NP, Tn - 1);
                    load_a_raw32(dl0 + (size_t)tn * U1 * 32, h, sa[s]);
#pragma unroll
                    for (int i = 0; i < KB; ++i) se[s][i] = enc[((size_t)b * Tn + tn) * J + kbc[i] * 32 + r];
                    cvt_a(af, row_ok, a);
                    frame(t, a, af, ev);
                }
            }
        }
    } else {
        for (int t = t_lo; t < t_end; ++t) {
            bf16x8 a[2];
            float af[2][8], ev[KB];
            load_a_frags(dlogits + (((size_t)b * Tn + t) * U1 + u0 + r) * ldl, row_ok, ldl, h, a, af);
#pragma unroll
            for (int i = 0; i < KB; ++i) ev[i] = f.kb[i] >= 0 ? ld1(enc + ((size_t)b * Tn + t) * J + f.kb[i] * 32 + r) : 0.f;
            frame(t, a, af, ev);
        }
    }
    if (denc_part && h == 0) {   // frames this tile never visited (beyond the utterance, or the tile lies outside the lattice)
        for (int t = max(t_end, t_lo); t < t_hi; ++t)
#pragma unroll
            for (int i = 0; i < KB; ++i)
                if (f.kb[i] >= 0) denc_part[(((size_t)ut * nb + b) * Tn + t) * J + f.kb[i] * 32 + r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        if (f.kb[i] < 0) continue;
        const int k = f.kb[i] * 32 + r;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int u = u0 + acc_row(g, h);
            const float dd = F16 ? dsum[i][g] : dacc[i][g];
            if (u < U1) {
                if (TS > 1) ddec_part[(((size_t)ts * nb + b) * U1 + u) * J + k] = dd;
                else st1(ddec + ((size_t)b * U1 + u) * J + k, dd);
            }
            slab_w[((size_t)((ts * nb + b) * nut + ut) * 32 + acc_row(g, h)) * J + k] = wacc[i][g];
        }
    }
    if (wslot == 0) {  // dbias partial: sum over the 32 lanes (u) that share h
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = bsum[s][j];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                if (r == 0) slab_b[(size_t)((ts * nb + b) * nut + ut) * 32 + 16 * s + 8 * h + j] = x;
            }
    }
}

// ---- Y: denc -----------------------------------------------------------------------------------
#define TG 8
template <typename T>
__global__ __launch_bounds__(256, 1) void joint_bwd_y_kernel(
    const float *__restrict__ dlogits, const T *__restrict__ enc, const T *__restrict__ dec,
    const float *__restrict__ W, T *__restrict__ denc, const int32_t *__restrict__ tlen,
    const int32_t *__restrict__ ulen, int Tn, int U1, int J, int V, int ldl, float slope) {
    const int b = blockIdx.z, t0 = blockIdx.x * TG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nslots = gridDim.y * 4, wslot = blockIdx.y * 4 + wave;
    const int Tb = tlen ? min(max(tlen[b], 1), Tn) : Tn;
    const int Ub = ulen ? min(max(ulen[b], 0), U1 - 1) : U1 - 1;
    BwdFrags f;
    load_w_frags(f, W, J, V, wslot, nslots, r, h);
    float eacc[TG][KB];
#pragma unroll
    for (int tt = 0; tt < TG; ++tt)
#pragma unroll
        for (int i = 0; i < KB; ++i) eacc[tt][i] = 0.f;
    const int nt = max(0, min(TG, Tb - t0));
#ifdef JY_PROFILE
    long long acc_t[4] = {0, 0, 0, 0}, t_prev = clock64();
#define JY_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long n_ = clock64(); acc_t[i] += n_ - t_prev; t_prev = n_; } while (0)
#else
#define JY_STAMP(i)
#endif
    if (nt > 0) {
        int kbc[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) kbc[i] = max(f.kb[i], 0);
        for (int u0 = 0; u0 <= Ub; u0 += 32) {
            float dv[KB][16];
            {   // dec values of the u tile: all requested together (clamped addresses), masked afterwards
                T raw[KB][16];
#pragma unroll
                for (int i = 0; i < KB; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) raw[i][g] = dec[((size_t)b * U1 + min(u0 + acc_row(g, h), U1 - 1)) * J + kbc[i] * 32 + r];
#pragma unroll
                for (int i = 0; i < KB; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) dv[i][g] = (f.kb[i] >= 0 && u0 + acc_row(g, h) < U1) ? (float)raw[i][g] : 0.f;
            }
            const bool row_ok = (u0 + r) < U1;
            JY_STAMP(0);   // dec values of the u tile
            if (ldl == 32) {
                // software pipeline over the TG frames: frame tt+1's dlogits row and enc values are in flight during frame tt
                const float *dl0 = dlogits + (((size_t)b * Tn) * U1 + min(u0 + r, U1 - 1)) * 32;
                float afn[2][8];
                T en[KB];
                load_a_raw32(dl0 + (size_t)t0 * U1 * 32, h, afn);
#pragma unroll
                for (int i = 0; i < KB; ++i) en[i] = enc[((size_t)b * Tn + t0) * J + kbc[i] * 32 + r];
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    float af[2][8], e[KB];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) af[s2][j] = afn[s2][j];
#pragma unroll
                    for (int i = 0; i < KB; ++i) e[i] = (float)en[i];
                    if (tt + 1 < TG) {
                        const int tn = min(t0 + tt + 1, Tn - 1);
                        load_a_raw32(dl0 + (size_t)tn * U1 * 32, h, afn);
#pragma unroll
                        for (int i = 0; i < KB; ++i) en[i] = enc[((size_t)b * Tn + tn) * J + kbc[i] * 32 + r];
                    }
                    bf16x8 a[2];
                    cvt_a(af, row_ok && tt < nt, a);
                    JY_STAMP(1);
#pragma unroll
                    for (int i = 0; i < KB; ++i) {
                        f32x16 D = {0};
                        D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f.wf[i][0], D, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], f.wf[i][1], D, 0, 0, 0);
                        float sacc = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) sacc += ((e[i] + dv[i][g]) > 0.f) ? D[g] : slope * D[g];
                        eacc[tt][i] += sacc;   // rows of dead frames / absent k-blocks are zero: a == 0 or wf == 0
                    }
                    JY_STAMP(2);
                }
            } else {
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    if (tt >= nt) continue;
                    const int t = t0 + tt;
                    bf16x8 a[2];
                    float af[2][8];
                    load_a_frags(dlogits + (((size_t)b * Tn + t) * U1 + u0 + r) * ldl, row_ok, ldl, h, a, af);
#pragma unroll
                    for (int i = 0; i < KB; ++i) {
                        if (f.kb[i] < 0) continue;
                        const float e = ld1(enc + ((size_t)b * Tn + t) * J + f.kb[i] * 32 + r);
                        f32x16 D = {0};
                        D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f.wf[i][0], D, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], f.wf[i][1], D, 0, 0, 0);
                        float sacc = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) sacc += ((e + dv[i][g]) > 0.f) ? D[g] : slope * D[g];
                        eacc[tt][i] += sacc;
                    }
                }
            }
        }
    }
#ifdef JY_PROFILE
    if (tid == 0 && blockIdx.x == 5 && blockIdx.y == 0 && b == 3) {
        long long *o = reinterpret_cast<long long *>(const_cast<T *>(dec));   // probe build only: clobbers 24 bytes of an input
        for (int i = 0; i < 3; ++i) o[i] = acc_t[i];
    }
#endif
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        const int t = t0 + tt;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const float x = eacc[tt][i] + __shfl_xor(eacc[tt][i], 32, 64);  // the two lane halves hold different u rows
            if (t < Tn && f.kb[i] >= 0 && h == 0) st1(denc + ((size_t)b * Tn + t) * J + f.kb[i] * 32 + r, x);
        }
    }
}

__global__ void joint_bwd_reduce_kernel(const float *__restrict__ slab_w, const float *__restrict__ slab_b,
                                        float *__restrict__ dW, float *__restrict__ dbias, int nslab, int J, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V * J) {
        const int v = i / J, k = i % J;
        float s = 0.f;
        for (int n = 0; n < nslab; ++n) s += slab_w[((size_t)n * 32 + v) * J + k];
        dW[i] = s;
    }
    if (i < V) {
        float s = 0.f;
        for (int n = 0; n < nslab; ++n) s += slab_b[(size_t)n * 32 + i];
        dbias[i] = s;
    }
}

// ============================================================================================
// loss
// ============================================================================================
struct RnntWs {
    float *lpb, *lpe_in, *lpe_out, *alpha, *beta, *lse, *logp, *xch;
    unsigned *err;   // split lattice only: set to 1 by a workgroup whose wait for its neighbour's edge value ran out (the costs are then NaN)
    int U1P, K;      // padded columns per frame; columns per lane of the ONE-wave layout
    int KT, NW, NC;  // the lattice kernel's plan: columns per thread, waves per workgroup, column blocks (workgroups) per (utterance, direction)
    int R, sh;       // rows per utterance plane; column -> skew shift (31: no skew)
};

static int rnnt_K(int U1) {
    int K = 1;
    while (64 * K < U1) K *= 2;
    return K;
}

// Lattice plan: columns per thread x waves x column blocks of rnnt_alphabeta_kernel, and whether the planes are stored SKEWED.
// A lattice step is an anti-diagonal: thread g (columns KT*g ..) works on frame t = s - g (alpha) resp. t + g = const (beta). Stored frame-major
// (row t), the 64 lanes of a wave touch 64 different rows per load / store - 64 cache lines per wave-instruction through the CU's one address
// path: at B = 1, T' = 4000, U = 1920 the four waves spent 1.36 us per step on six such instructions (5.8 ms per direction, "17 GB/s"). Stored at
// row t + g(u) instead (g(u) = u / KT; G - 1 more rows per utterance), a step's operands and results are ONE contiguous row for the whole
// lattice, in both directions. The layout is private to this file: rnnt_lp writes it, rnnt_grad reads it through ws_at().
//   U1 <= 256 (the benchmark's U = 120): one wave per lattice, frame-major planes (skewing them costs rnnt_lp / rnnt_grad their coalescing:
//            +10 - 20 us per step at configs[1], measured);
//   longer targets: skewed planes, up to 8 waves in one workgroup per lattice (2.5 ms at B = 1, T' = 4000, U = 1920: the CU's VALUs, ~0.55 us per
//            step for 2048 columns); a single long utterance (2 B NC <= 32 workgroups): one single-wave workgroup per block of 128 columns
//            instead (rnnt_alphabeta_kernel MC: 1.9 ms; at B = 8, T' = 1000, U = 400 the 8-wave form was faster, 0.72 against 0.75 ms).
// tsasr_rnnt_lattice_plan() overrides the choice (tests run all three forms against each other, bit for bit; tools/rnnt_bench.py).
static int g_plan_skew = -1, g_plan_waves = -1, g_plan_mc = -1;      // -1: default
static void rnnt_plan(int B, int U1, int *KT, int *NW, int *NC, int *skew) {
    const int max_waves = g_plan_waves > 0 ? g_plan_waves : 8;
    const int skew_mode = g_plan_skew >= 0 ? g_plan_skew : 1;        // 0 never, 1 when the lattice runs on more than one wave, 2 always
    const int mc_cols = g_plan_mc >= 0 ? g_plan_mc : 2;              // 0: never split a lattice over workgroups; 1, 2, 4: columns per thread of a split one
    const int mc_limit = g_plan_mc > 0 ? 2048 : 32;                  // (forced: whatever still fits on the chip at once)
    const int K = rnnt_K(U1);
    *NC = 1;
    if (K >= 8 && skew_mode != 0 && (mc_cols == 1 || mc_cols == 2 || mc_cols == 4) && (long long)2 * B * (K / mc_cols) <= mc_limit) {
        *KT = mc_cols; *NW = 1; *NC = K / mc_cols; *skew = 1;
        return;
    }
    int nw = 1;
    if (K >= 8) {                                  // at least 2 columns per thread
        nw = std::min(max_waves, K / 2);
        while (nw & (nw - 1)) nw &= nw - 1;        // power of two
    }
    *KT = K / nw;
    *NW = nw;
    *skew = skew_mode == 2 || (skew_mode == 1 && nw > 1);
}

static size_t rnnt_plane_floats(int B, int T, int U1) {
    int kt, nw, nc, skew;
    rnnt_plan(B, U1, &kt, &nw, &nc, &skew);
    return (size_t)B * (T + (skew ? 64 * nw * nc - 1 : 0)) * 64 * rnnt_K(U1);
}
static size_t rnnt_xch_floats(int B, int T, int U1) {
    int kt, nw, nc, skew;
    rnnt_plan(B, U1, &kt, &nw, &nc, &skew);
    return nc > 1 ? (size_t)2 * B * nc * (T + 1) + 64 : 0;      // (+ 64: the split lattice's error word, see AB_SPIN_LIMIT)
}

static RnntWs rnnt_carve(void *ws, int B, int T, int U1) {
    RnntWs w;
    int skew;
    w.K = rnnt_K(U1);
    w.U1P = 64 * w.K;
    rnnt_plan(B, U1, &w.KT, &w.NW, &w.NC, &skew);
    w.R = T + (skew ? 64 * w.NW * w.NC - 1 : 0);
    w.sh = 31;
    if (skew) { w.sh = 0; while ((1 << w.sh) < w.KT) ++w.sh; }
    const size_t n = rnnt_plane_floats(B, T, U1);
    float *p = reinterpret_cast<float *>(ws);
    w.lpb = p; w.lpe_in = p + n; w.lpe_out = p + 2 * n; w.alpha = p + 3 * n; w.beta = p + 4 * n; w.lse = p + 5 * n;
    w.logp = p + 6 * n;
    w.xch = w.logp + align_up((size_t)B, 64);
    w.err = w.NC > 1 ? reinterpret_cast<unsigned *>(w.xch + rnnt_xch_floats(B, T, U1) - 64) : nullptr;
    return w;
}

// element (b, t, u) of a lattice plane
__device__ __forceinline__ size_t ws_at(const RnntWs &w, int b, int t, int u) { return ((size_t)b * w.R + t + (u >> w.sh)) * w.U1P + u; }

// log(e^a + e^b) for a, b finite or -inf (never NaN / +inf): the lattice recursion's one operation, ~10 dependent instructions.
// Both -inf: min - max is NaN and the clamp (IEEE maxNum) returns -200, so e = 0 and the result is m = -inf without a compare / select;
// d < -200 underflows to e = 0 either way. v_exp_f32 / v_log_f32 directly: 1 + e lies in [1, 2], where the denormal pre-scaling and the
// split-constant correction of the library logf (14 more instructions per cell in the ISA of round 3) have nothing to do.
__device__ __forceinline__ float logaddexp_f(float a, float b) {
    const float m = fmaxf(a, b);
    const float d = fmaxf(fminf(a, b) - m, -200.f);
    return fmaf(__builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d * 1.44269504089f)), 0.69314718056f, m);
}

// (1) fused log-softmax per lattice cell: 8 lanes per cell, float4 per lane and chunk
__global__ __launch_bounds__(256) void rnnt_lp_kernel(const float *__restrict__ logits, const int32_t *__restrict__ targets,
                                                      int ldt, const int32_t *__restrict__ tlen,
                                                      const int32_t *__restrict__ ulen, RnntWs w, int B, int Tn, int U1,
                                                      int V, int ldl, int blank) {
    const int sub = threadIdx.x & 7;
    const long long nrows = (long long)B * Tn * U1;
    if (V <= 32) {
        // the benchmark's case (V = 29): ONE round trip per cell - lengths, the cell's row (one float4 per lane) and its label are requested
        // together, blank / label log-probabilities are picked out of the lanes' registers (a select + the same three DPP steps as the sums).
        // Round 5: an 8-lane group walks LP_R cells (32 rows apart: a workgroup's loads stay contiguous 4 KB pieces) with all their requests
        // in flight before the first one is used - one float4 per thread in flight left the kernel at 2.5 - 3 TB/s.
        constexpr int LP_R = 4;
        const long long row0 = (long long)blockIdx.x * (32 * LP_R) + (threadIdx.x >> 3);
        const int c = sub * 4;
        float4 x[LP_R];
        int lab[LP_R], tl[LP_R], ul[LP_R], uu[LP_R], tt[LP_R], bb[LP_R];
        bool in[LP_R];
#pragma unroll
        for (int k = 0; k < LP_R; ++k) {
            const long long row = row0 + 32 * k;
            in[k] = row < nrows;
            const long long rc = in[k] ? row : nrows - 1;
            uu[k] = (int)(rc % U1); tt[k] = (int)((rc / U1) % Tn); bb[k] = (int)(rc / ((long long)U1 * Tn));
            x[k] = *reinterpret_cast<const float4 *>(logits + rc * ldl + min(c, ldl - 4));
            tl[k] = tlen[bb[k]]; ul[k] = ulen[bb[k]];
            lab[k] = targets[(size_t)bb[k] * ldt + min(uu[k], max(ldt - 1, 0))];
        }
#pragma unroll
        for (int k = 0; k < LP_R; ++k) {
            const int b = bb[k], t = tt[k], u = uu[k];
            const int Tb = min(max(tl[k], 1), Tn), Ub = min(max(ul[k], 0), U1 - 1);
            const bool live = in[k] && t < Tb && u <= Ub;      // (uniform over the 8-lane group; the DPP steps below run for every lane)
            const float xs[4] = {c + 0 < V ? x[k].x : NEG_INF, c + 1 < V ? x[k].y : NEG_INF, c + 2 < V ? x[k].z : NEG_INF, c + 3 < V ? x[k].w : NEG_INF};
            float m = fmaxf(fmaxf(xs[0], xs[1]), fmaxf(xs[2], xs[3]));
            m = fmaxf(m, dpp_mov<0xB1>(m));
            m = fmaxf(m, dpp_mov<0x4E>(m));
            m = fmaxf(m, dpp_mov<0x141>(m));
            const int kl = min(max(lab[k], 0), V - 1);
            float sm = 0.f, vb = 0.f, vl = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c + j < V) sm += __expf(xs[j] - m);
                vb += (c + j == blank) ? xs[j] : 0.f;
                vl += (c + j == kl) ? xs[j] : 0.f;
            }
            sm += dpp_mov<0xB1>(sm);   vb += dpp_mov<0xB1>(vb);   vl += dpp_mov<0xB1>(vl);
            sm += dpp_mov<0x4E>(sm);   vb += dpp_mov<0x4E>(vb);   vl += dpp_mov<0x4E>(vl);
            sm += dpp_mov<0x141>(sm);  vb += dpp_mov<0x141>(vb);  vl += dpp_mov<0x141>(vl);
            if (sub == 0 && live) {
                const float lse = m + __logf(sm);
                const size_t o = ws_at(w, b, t, u);
                w.lse[o] = lse;
                w.lpb[o] = vb - lse;
                if (u < Ub) {
                    const float e = vl - lse;
                    w.lpe_out[o] = e;
                    w.lpe_in[ws_at(w, b, t, u + 1)] = e;
                }
            }
        }
        return;
    }
    const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (row >= nrows) return;  // whole 8-lane groups leave together
    const int u = row % U1, t = (row / U1) % Tn, b = row / ((long long)U1 * Tn);
    const float *rowp = logits + row * ldl;
    const int Tb = min(max(tlen[b], 1), Tn), Ub = min(max(ulen[b], 0), U1 - 1);
    if (t >= Tb || u > Ub) return;
    float m = NEG_INF;
    for (int c = sub * 4; c < V; c += 32) {
        const float4 x = *reinterpret_cast<const float4 *>(rowp + c);
        m = fmaxf(m, x.x);
        if (c + 1 < V) m = fmaxf(m, x.y);
        if (c + 2 < V) m = fmaxf(m, x.z);
        if (c + 3 < V) m = fmaxf(m, x.w);
    }
    // 8 lanes per lattice cell: reductions stay inside the 16-lane DPP rows (quad swaps, then the half-row mirror) - no LDS crossbar
    m = fmaxf(m, dpp_mov<0xB1>(m));
    m = fmaxf(m, dpp_mov<0x4E>(m));
    m = fmaxf(m, dpp_mov<0x141>(m));
    float s = 0.f;
    for (int c = sub * 4; c < V; c += 32) {
        const float4 x = *reinterpret_cast<const float4 *>(rowp + c);
        s += __expf(x.x - m);
        if (c + 1 < V) s += __expf(x.y - m);
        if (c + 2 < V) s += __expf(x.z - m);
        if (c + 3 < V) s += __expf(x.w - m);
    }
    s += dpp_mov<0xB1>(s);
    s += dpp_mov<0x4E>(s);
    s += dpp_mov<0x141>(s);
    if (sub == 0) {
        const float lse = m + __logf(s);
        const size_t o = ws_at(w, b, t, u);
        w.lse[o] = lse;
        w.lpb[o] = rowp[blank] - lse;
        if (u < Ub) {
            const float e = rowp[min(max(targets[(size_t)b * ldt + u], 0), V - 1)] - lse;
            w.lpe_out[o] = e;
            w.lpe_in[ws_at(w, b, t, u + 1)] = e;
        }
    }
}

// (2) alpha / beta: one workgroup of NW waves per (utterance, direction), blockIdx.x = 2b + dir. Thread g = wave * 64 + lane owns the K
// columns u = K*g .. K*g+K-1 and is skewed by g frames (alpha: t = s - g at step s), so a step needs only the value the neighbouring
// thread produced in the step before: inside a wave by a DPP wave shift, across a wave boundary through one LDS word per wave
// (double-buffered by step parity) behind the step's workgroup barrier. NW = 1 (short targets: the benchmark's U = 120 is K = 2) has
// neither LDS nor barriers; long targets spread over up to 16 waves (U = 1920: 16 waves x 64 lanes x 2 columns) instead of 32
// sequentially dependent columns per lane in ONE wave (49.7 ms of an 85 ms long-form step before).
// end of a lattice step: the boundary words in LDS are visible to the other waves. NOT __syncthreads(): that also waits for every
// outstanding global load (s_waitcnt vmcnt(0)) - here the operands of the next eight steps, requested on purpose - and made a step
// cost a full memory round trip (1.5 - 3 us per step with 4 - 16 waves)
__device__ __forceinline__ void ab_step_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// MC ("multi-CU", long targets in small batches: one CU's VALUs bound the lattice at ~0.55 us per step for 2048 columns): the columns of ONE
// lattice are cut into NC blocks of 64*K, one single-wave workgroup each - no LDS, no barrier - pipelined along the skew: block c runs the
// same steps 64*c later. The one value per step that crosses a block boundary (alpha(t, last column of block c-1) resp. beta(t, first column
// of block c+1)) goes through a global array `xch` [pair][block][t] that the host fills with NaN bits before the launch: the producer's edge
// lane writes it with an agent-scope store, the consumer fetches 64 frames at a time, 16 steps before it needs them, and spins only while a
// NaN is left among the frames it is about to use (a lattice value is never NaN). Workgroup ids are stage-major, and a stage waits only for
// the stage before it, which the dispatcher has started earlier: no deadlock; a bounded spin: no hang (the costs come out NaN instead).
constexpr int AB_SPIN_LIMIT = 1 << 16;      // x ~3 us per poll: 0.2 s per block at worst

__device__ __forceinline__ unsigned xch_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xch_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The waiting path's loads (same cache policy as xch_load: sc1, served at device scope) as asm that is complete when it returns: hipcc does not
// see a memory operation in it. With ordinary loads in that conditional path, the join behind it left the compiler unsure how many operations
// are in flight in EVERY later step, and its counted waits for the prefetched operands came out as vmcnt(7) ... vmcnt(0): the window drained
// per step. (base: wave-uniform; off: byte offsets)
__device__ __forceinline__ void xch_poll2(const unsigned *base, unsigned off_a, unsigned off_b, unsigned &a, unsigned &b) {
    asm volatile("global_load_dword %0, %2, %4 sc1\n\tglobal_load_dword %1, %3, %4 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(off_a), "v"(off_b), "s"(base) : "memory");
}
__device__ __forceinline__ void xch_load4_sync(const unsigned *base, const unsigned *off, unsigned *out) {
    asm volatile("global_load_dword %0, %4, %8 sc1\n\tglobal_load_dword %1, %5, %8 sc1\n\tglobal_load_dword %2, %6, %8 sc1\n\t"
                 "global_load_dword %3, %7, %8 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base)
                 : "memory");
}

template <int K, int NW, bool SKEW, bool MC>
__global__ __launch_bounds__(64 * NW) void rnnt_alphabeta_kernel(RnntWs w, const int32_t *__restrict__ tlen,
                                                                 const int32_t *__restrict__ ulen, float *__restrict__ costs,
                                                                 int Tn, int U1) {
    static_assert(!MC || (NW == 1 && SKEW), "multi-CU lattices: one wave per column block, skewed planes");
    constexpr int G = 64 * NW;
    __shared__ float edge_lds[2][NW > 1 ? NW : 1];
    const int NC = MC ? w.NC : 1, npairs = gridDim.x / NC;
    const int pair = MC ? blockIdx.x % npairs : blockIdx.x, stage = MC ? blockIdx.x / npairs : 0;
    const int b = pair >> 1, dir = pair & 1, g = threadIdx.x, l = g & 63, wave = g >> 6;
    const int cb = MC ? (dir == 0 ? stage : NC - 1 - stage) : 0;        // column block; the pipeline runs left to right for alpha, right to left for beta
    const int gc = cb * 64 + g;                                          // thread index within the lattice: owns columns K*gc ..
    const int U1P = w.U1P;
    const int Tb = min(max(tlen[b], 1), Tn), Ub = min(max(ulen[b], 0), U1 - 1);
    if (MC && cb * 64 * K > Ub) return;                                  // a block wholly beyond the last label: nothing in it is read
    const size_t base = (size_t)b * w.R * U1P + (size_t)K * gc;
    const int nsteps = Tb + G - 1;
    const int rmax = w.R - 1;     // SKEW: frame t of this thread's columns is row t + gc, and a step touches ONE row for the whole workgroup
    const int roff = cb * 64;     // ... = local step + roff
    float prev[K], cur[K];
#pragma unroll
    for (int i = 0; i < K; ++i) prev[i] = cur[i] = NEG_INF;
    float edge = NEG_INF;  // boundary value handed to the neighbour thread
    // MC: the neighbour block's edge value is one more per-step operand, requested PF steps ahead like the log-probabilities (every lane
    // asks for the edge lane's frame: one address per wave-instruction) - the steps stay one straight line with counted waits. A value that
    // has not arrived (NaN bits) sends the wave into xch_wait: it sleeps until the neighbour is XCH_LEAD steps ahead, then asks again for the
    // whole window; after that the pipeline runs with that lead and the wait is not entered again.
    // (rows of Tn + 1 words: the last one takes the edge lane's stores from steps outside its frames, so that the store needs no guard)
    unsigned *xch_out = nullptr;
    const unsigned *xch_in = nullptr;
    if (MC) {
        unsigned *x = reinterpret_cast<unsigned *>(w.xch) + ((size_t)pair * NC) * (Tn + 1);
        const int up = dir == 0 ? cb - 1 : cb + 1;                      // the block this one waits for
        xch_out = x + (size_t)cb * (Tn + 1);                            // (nobody reads the row of a block without a successor)
        xch_in = xch_out;
        if (up >= 0 && up < NC && up * 64 * K <= Ub) xch_in = x + (size_t)up * (Tn + 1);
    }
    const bool has_in = MC && xch_in != xch_out;
    // frame of the consuming edge lane (alpha: lane 0, beta: lane 63) at local step ss, clamped into the utterance
    auto xch_idx = [&](int ss) { return min(max(dir == 0 ? ss : Tb - 1 - ss, 0), Tb - 1); };
    auto xch_ptr = [&](int ss) { return xch_in + xch_idx(ss); };
    // The per-step operands (log-probabilities of blank / label for this thread's K columns at its current frame) are requested PF
    // steps ahead, unconditionally (frame index clamped): fetched inside the guarded step they cost one L2 round trip per step,
    // and the steps are strictly sequential (310 us for 313 steps before; the arithmetic of a step is ~50 cycles).
    constexpr int PF = 8, XCH_LEAD = PF + 16;      // (MC: 5 memory operations per step, vmcnt counts to 63; PF = 12 measured equal)
    // The window is filled by PF "null steps" s = -PF .. -1 of the ramp-up loop itself (no thread has a frame there; they only request the
    // operands of steps 0 .. PF-1), not by a prologue of loads: hipcc counts its s_waitcnt vmcnt(N) from the state at the loop's entry, and
    // behind a prologue of bare loads (no stores between them) every step's first wait allowed 15 operations in flight instead of the 22 (40
    // with the exchange) a step really has behind its operands - the loads of the last 3 steps had to land, not those of 8 steps ago.
    float qb[PF][K], qe[PF][K];
    unsigned qx[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        qx[d] = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i) qb[d][i] = qe[d][i] = 0.f;
    }
    // step s0 + d waits for its edge value; on return every slot of the window has been requested again (slot j holds step s0 + j for
    // j >= d, step s0 + PF + j for j < d: those were refilled by the steps before this one)
    auto xch_wait = [&](int s0, int d) {
        static_assert(PF % 4 == 0, "xch_load4_sync");
        int spins = 0;
        unsigned v, vf;
        do {
            __builtin_amdgcn_s_sleep(16);
            xch_poll2(xch_in, 4u * xch_idx(s0 + d), 4u * xch_idx(s0 + d + XCH_LEAD), v, vf);
        } while ((v == 0xffffffffu || vf == 0xffffffffu) && ++spins < AB_SPIN_LIMIT);
        if (spins >= AB_SPIN_LIMIT && (threadIdx.x & 63) == 0) *w.err = 1u;      // the producer never came (not co-resident?): the host reads this word (tsasr_rnnt_loss_error_word_offset)
        unsigned off[PF], out[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) off[j] = 4u * xch_idx(j >= d ? s0 + j : s0 + PF + j);
#pragma unroll
        for (int j = 0; j < PF; j += 4) xch_load4_sync(xch_in, off + j, out + j);
#pragma unroll
        for (int j = 0; j < PF; ++j) qx[j] = out[j];
    };
    if (dir == 0) {
        auto fetch = [&](int ss, float (&vb)[K], float (&ve)[K]) {
            const int t = min(max(ss - g, 0), Tb - 1);
            // SKEW: rows ss - 1 and ss whatever the thread's frame is (outside its frames the values are not used)
            const float *pb = w.lpb + base + (size_t)(SKEW ? min(max(ss + roff - 1, 0), rmax) : (t > 0 ? t - 1 : 0)) * U1P;
            const float *pe = w.lpe_in + base + (size_t)(SKEW ? min(ss + roff, rmax) : t) * U1P;
#pragma unroll
            for (int i = 0; i < K; ++i) { vb[i] = pb[i]; ve[i] = pe[i]; }
        };
        // a chunk of PF steps; GUARD = false when every thread has a valid frame in every step of the chunk (G-1 <= s < Tb). The guarded
        // form is STRAIGHT-LINE code too: a thread outside its frames computes on a clamped frame and keeps its old values by select, and
        // stores them again at the clamped frame (before its first frame: -inf into alpha(0, u), overwritten in order by the same thread
        // when it gets there; after its last: alpha(Tb-1, u) once more). With the step inside a divergent guard the compiler drained
        // every outstanding prefetch behind it (37 x s_waitcnt vmcnt(0)): the 2 x 63 ramp steps of a 313-step lattice cost a memory
        // round trip each - 0.6 us against 0.2 us for an unguarded step.
        auto chunk = [&](int s0, auto guard_tag) {
            constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int s = s0 + d;
                const int t = s - g;
                const bool valid = !GUARD || (t >= 0 && t < Tb);
                const int tc = GUARD ? min(max(t, 0), Tb - 1) : t;
                float left = __builtin_amdgcn_update_dpp(NEG_INF, edge, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (NW > 1 && l == 0 && wave > 0) left = edge_lds[(s + 1) & 1][wave - 1];          // written in step s-1
                if (MC) {
                    unsigned xs = __builtin_amdgcn_readfirstlane(qx[d]);
                    if (has_in && xs == 0xffffffffu && s < Tb) {
                        xch_wait(s0, d);
                        xs = __builtin_amdgcn_readfirstlane(qx[d]);
                    }
                    left = (has_in && l == 0) ? __uint_as_float(xs) : left;
                }
                // Off the dependent chain: the no-emit operand (alpha(0,0) = 0 enters as "no-emit = 0, emit = -inf") and the emit
                // log-probability with the lattice's edges folded in as -inf operands (columns beyond Ub, column 0: both operands -inf give
                // -inf). On the chain per column: one add and logaddexp_f. Straight-line: no branch around a cell.
                float ne[K], qm[K], nv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const int u = K * gc + i;
                    ne[i] = (t > 0) ? prev[i] + qb[d][i] : (u == 0 ? 0.f : NEG_INF);
                    qm[i] = (u > 0) ? qe[d][i] : NEG_INF;
                    if (u > Ub) ne[i] = qm[i] = NEG_INF;
                }
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float lft = (i == 0) ? left : nv[i > 0 ? i - 1 : 0];
                    nv[i] = logaddexp_f(ne[i], lft + qm[i]);
                }
                // SKEW: outside its frames a thread writes the slot of a frame that does not exist (row s, its own columns) or, in the
                // clamped rows past the last step, its unchanged last value over itself
                float *pa = w.alpha + base + (size_t)(SKEW ? min(max(s + roff, 0), rmax) : tc) * U1P;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    cur[i] = valid ? nv[i] : cur[i];
                    pa[i] = cur[i];
                    prev[i] = cur[i];
                }
                edge = cur[K - 1];
                if (NW > 1 && l == 63) edge_lds[s & 1][wave] = edge;
                if (MC) {     // every lane stores the edge lane's value to the edge lane's slot: one dword, no branch
                    const int te = s - 63;
                    xch_store(xch_out + ((te >= 0 && te < Tb) ? te : Tn), __builtin_amdgcn_readlane(__float_as_uint(edge), 63));
                }
                if (MC) qx[d] = xch_load(xch_ptr(s + PF));
                fetch(s + PF, qb[d], qe[d]);
                if (NW > 1) ab_step_barrier();
            }
        };
        // Ramp-up, steady state (every thread on a valid frame: s0 >= G-1 and s0 + PF < Tb), ramp-down as THREE loops of one body each (a
        // last partial chunk runs to its end: the steps past nsteps hold no valid frame). As one loop choosing between the two forms per
        // chunk, the join in front of the back-edge copied the whole prefetch window between the two forms' registers - 32 v_mov_b64 behind an
        // s_waitcnt vmcnt(0) that drained the loads just requested: one memory round trip per 8 steps (~0.2 - 0.3 us per step at T' = 4000).
        int s0 = -PF;
        for (; s0 < nsteps && !(s0 >= G - 1 && s0 + PF < Tb); s0 += PF) chunk(s0, std::true_type{});
        for (; s0 < nsteps && s0 + PF < Tb; s0 += PF) chunk(s0, std::false_type{});
        for (; s0 < nsteps; s0 += PF) chunk(s0, std::true_type{});
        // the thread that owns column Ub still holds alpha(Tb-1, Ub)
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (K * gc + i == Ub) {
                const float lp = cur[i] + w.lpb[base + (size_t)(Tb - 1 + (SKEW ? gc : 0)) * U1P + i];
                w.logp[b] = lp;
                costs[b] = -lp;
            }
    } else {
        auto fetch = [&](int ss, float (&vb)[K], float (&ve)[K]) {
            const int t = min(max(Tb - 1 - (ss - (G - 1 - g)), 0), Tb - 1);
            const int row = SKEW ? min(max(Tb + G - 2 - ss + roff, 0), rmax) : t;      // t + gc is the same for every thread of a step
            const float *pb = w.lpb + base + (size_t)row * U1P;
            const float *pe = w.lpe_out + base + (size_t)row * U1P;
#pragma unroll
            for (int i = 0; i < K; ++i) { vb[i] = pb[i]; ve[i] = pe[i]; }
        };
        auto chunk = [&](int s0, auto guard_tag) {       // straight-line in both forms, as in the alpha direction
            constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int s = s0 + d;
                const int t = Tb - 1 - (s - (G - 1 - g));
                const bool valid = !GUARD || (t >= 0 && t < Tb);
                const int tc = GUARD ? min(max(t, 0), Tb - 1) : t;
                float right = __builtin_amdgcn_update_dpp(NEG_INF, edge, 0x130, 0xf, 0xf, false);  // wave_shl:1
                if (NW > 1 && l == 63 && wave < NW - 1) right = edge_lds[(s + 1) & 1][wave + 1];
                if (MC) {
                    unsigned xs = __builtin_amdgcn_readfirstlane(qx[d]);
                    if (has_in && xs == 0xffffffffu && s < Tb) {
                        xch_wait(s0, d);
                        xs = __builtin_amdgcn_readfirstlane(qx[d]);
                    }
                    right = (has_in && l == 63) ? __uint_as_float(xs) : right;
                }
                float ne[K], qm[K], nv[K];     // as in the alpha direction; beta(Tb-1, Ub) = lp_blank enters as "no-emit = lp_blank, emit = -inf"
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const int u = K * gc + i;
                    const float lb = qb[d][i];
                    ne[i] = (t < Tb - 1) ? prev[i] + lb : (u == Ub ? lb : NEG_INF);
                    qm[i] = (u < Ub) ? qe[d][i] : NEG_INF;
                    if (u > Ub) ne[i] = NEG_INF;
                }
#pragma unroll
                for (int i = K - 1; i >= 0; --i) {
                    const float rgt = (i == K - 1) ? right : nv[i < K - 1 ? i + 1 : K - 1];
                    nv[i] = logaddexp_f(ne[i], rgt + qm[i]);
                }
                float *pbeta = w.beta + base + (size_t)(SKEW ? min(max(Tb + G - 2 - s + roff, 0), rmax) : tc) * U1P;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    cur[i] = valid ? nv[i] : cur[i];
                    pbeta[i] = cur[i];
                    prev[i] = cur[i];
                }
                edge = cur[0];
                if (NW > 1 && l == 0) edge_lds[s & 1][wave] = edge;
                if (MC) {
                    const int te = Tb - 1 - (s - 63);
                    xch_store(xch_out + ((te >= 0 && te < Tb) ? te : Tn), __builtin_amdgcn_readlane(__float_as_uint(edge), 0));
                }
                if (MC) qx[d] = xch_load(xch_ptr(s + PF));
                fetch(s + PF, qb[d], qe[d]);
                if (NW > 1) ab_step_barrier();
            }
        };
        int s0 = -PF;
        for (; s0 < nsteps && !(s0 >= G - 1 && s0 + PF < Tb); s0 += PF) chunk(s0, std::true_type{});
        for (; s0 < nsteps && s0 + PF < Tb; s0 += PF) chunk(s0, std::false_type{});
        for (; s0 < nsteps; s0 += PF) chunk(s0, std::true_type{});
    }
}

// (3) gradient w.r.t. logits; covers EVERY row of dlogits (zeros outside the lattice)
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const float *__restrict__ logits, const int32_t *__restrict__ targets,
                                                        int ldt, const int32_t *__restrict__ tlen,
                                                        const int32_t *__restrict__ ulen, const float *__restrict__ gscale,
                                                        float *__restrict__ dlogits, RnntWs w, int B, int Tn, int U1, int V,
                                                        int ldl, int blank) {
    const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    if (row >= (long long)B * Tn * U1) return;
    const int u = row % U1, t = (row / U1) % Tn, b = row / ((long long)U1 * Tn);
    const int Tb = min(max(tlen[b], 1), Tn), Ub = min(max(ulen[b], 0), U1 - 1);
    float *orow = dlogits + row * ldl;
    if (t >= Tb || u > Ub) {
        for (int c = sub * 4; c < ldl; c += 32) *reinterpret_cast<float4 *>(orow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const size_t o = ws_at(w, b, t, u);
    const float logp = w.logp[b], a = w.alpha[o], be = w.beta[o], lse = w.lse[o], lpb = w.lpb[o];
    const float gs = gscale[b];
    const float c0 = a + be - logp - lse;
    float gblank;
    if (t < Tb - 1) gblank = __expf(a + lpb + w.beta[o + w.U1P] - logp);
    else gblank = (u == Ub) ? __expf(a + lpb - logp) : 0.f;
    int lab = -1;
    float gemit = 0.f;
    if (u < Ub) {
        lab = min(max(targets[(size_t)b * ldt + u], 0), V - 1);
        gemit = __expf(a + w.lpe_out[o] + w.beta[ws_at(w, b, t, u + 1)] - logp);
    }
    const float *rowp = logits + row * ldl;
    for (int c = sub * 4; c < ldl; c += 32) {
        const float4 x = *reinterpret_cast<const float4 *>(rowp + c);
        float g[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = c + j;
            float val = 0.f;
            if (v < V) {
                val = __expf(c0 + g[j]);
                if (v == blank) val -= gblank;
                if (v == lab) val -= gemit;
            }
            g[j] = val * gs;
        }
        *reinterpret_cast<float4 *>(orow + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

template <int K, int NW>
static void launch_ab(RnntWs w, const int32_t *tlen, const int32_t *ulen, float *costs, int B, int T, int U1, hipStream_t st) {
    if (w.sh != 31) rnnt_alphabeta_kernel<K, NW, true, false><<<2 * B, 64 * NW, 0, st>>>(w, tlen, ulen, costs, T, U1);
    else rnnt_alphabeta_kernel<K, NW, false, false><<<2 * B, 64 * NW, 0, st>>>(w, tlen, ulen, costs, T, U1);
}
template <int K>
static int launch_ab_mc(RnntWs w, const int32_t *tlen, const int32_t *ulen, float *costs, int B, int T, int U1, hipStream_t st) {
    // every exchange word "not there yet" (NaN bits) before the first workgroup starts
    if (hipMemsetAsync(w.xch, 0xff, rnnt_xch_floats(B, T, U1) * sizeof(float), st) != hipSuccess) {
        tsasr_set_error("tsasr_rnnt_loss_fwd: clearing the lattice exchange buffer failed");
        return TSASR_E_LAUNCH;
    }
    rnnt_alphabeta_kernel<K, 1, true, true><<<2 * B * w.NC, 64, 0, st>>>(w, tlen, ulen, costs, T, U1);
    return 0;
}

// (columns per thread, waves, blocks) from rnnt_plan. Measured at B = 1, T' = 4000, U = 1920 with frame-major planes (round 2): 1 wave 50.5 ms,
// 2 waves 12.7, 4 waves 6.2, 8 waves 8.9, 16 waves 15.3 - every added wave brought its own 64-line loads and stores; skewed: profiles/r04_notes.md.
static int launch_alphabeta(RnntWs w, const int32_t *tlen, const int32_t *ulen, float *costs, int B, int T, int U1, hipStream_t st) {
    if (w.NC > 1) {
        if (w.KT == 1) return launch_ab_mc<1>(w, tlen, ulen, costs, B, T, U1, st);
        if (w.KT == 2) return launch_ab_mc<2>(w, tlen, ulen, costs, B, T, U1, st);
        if (w.KT == 4) return launch_ab_mc<4>(w, tlen, ulen, costs, B, T, U1, st);
    }
#define AB_CASE(K_, NW_) if (w.NC == 1 && w.KT == K_ && w.NW == NW_) { launch_ab<K_, NW_>(w, tlen, ulen, costs, B, T, U1, st); return 0; }
    AB_CASE(1, 1) AB_CASE(2, 1) AB_CASE(4, 1) AB_CASE(8, 1) AB_CASE(16, 1) AB_CASE(32, 1)
    AB_CASE(4, 2) AB_CASE(8, 2) AB_CASE(16, 2)
    AB_CASE(2, 4) AB_CASE(4, 4) AB_CASE(8, 4)
    AB_CASE(2, 8) AB_CASE(4, 8)
    AB_CASE(2, 16)
#undef AB_CASE
    tsasr_set_error("tsasr_rnnt_loss_fwd: no lattice kernel for %d columns per thread x %d waves x %d blocks", w.KT, w.NW, w.NC);
    return TSASR_E_INVALID;
}

// ============================================================================================
// C-ABI
// ============================================================================================
extern "C" {

int tsasr_joint_fwd(const void *enc, const void *dec, const float *W, const float *bias, float *logits, int B, int T,
                    int U1, int J, int V, int ldl, int io_dtype, float slope, void *stream) {
    TSASR_CHECK_ARG(enc && dec && W && bias && logits, "tsasr_joint_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0, "tsasr_joint_fwd: empty batch (B=%d T=%d U1=%d)", B, T, U1);
    TSASR_CHECK_ARG(J > 0 && J % 16 == 0, "tsasr_joint_fwd: J=%d must be a multiple of 16", J);
    TSASR_CHECK_ARG(V > 0 && V <= 32, "tsasr_joint_fwd: V=%d not supported (1..32)", V);
    TSASR_CHECK_ARG(ldl >= V && ldl % 4 == 0 && ldl <= 32, "tsasr_joint_fwd: ldl=%d must be a multiple of 4 in [V,32]", ldl);
    const size_t esz = io_dtype == TSASR_F32 ? sizeof(float) : sizeof(bf16_t);
    const size_t lds = (size_t)32 * lds_stride(J) * (sizeof(bf16_t) + esz) + (size_t)4 * J * esz;   // W tile, dec tile, one enc row per wave
    TSASR_CHECK_ARG((size_t)J * esz <= 3 * 64 * 16, "tsasr_joint_fwd: J=%d too wide for the per-wave row staging", J);
    TSASR_CHECK_ARG(lds <= 160 * 1024, "tsasr_joint_fwd: J=%d needs %zu B of LDS (>160 KiB)", J, lds);
    const int nut = cdiv(U1, 32);
    int tsplit = 1;  // enough workgroups to cover 256 CUs twice, at least 8 frames per wave
    while (nut * B * tsplit < 512 && T / (tsplit * 2) >= 32) tsplit *= 2;
    dim3 grid(nut, tsplit, B);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == TSASR_F32) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)joint_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        joint_fwd_kernel<float><<<grid, 256, lds, st>>>((const float *)enc, (const float *)dec, W, bias, logits, T, U1, J, V, ldl, slope);
    } else if (io_dtype == TSASR_BF16 && J == 640) {      // the TS-ASR joint (joint_dim 640): head matrix in registers, two workgroups per CU
        const size_t lds_r = (size_t)32 * (J + 8) * sizeof(bf16_t) + (size_t)4 * J * sizeof(bf16_t) + 32 * sizeof(float);
        while (nut * B * tsplit < 1024 && T / (tsplit * 2) >= 16) tsplit *= 2;      // (512 / 1024 / 2048 workgroups measured equal)
        if (slope >= 0.f && slope <= 1.f)
            joint_fwd_regw_kernel<40, true><<<dim3(nut, tsplit, B), 256, lds_r, st>>>((const bf16_t *)enc, (const bf16_t *)dec, W, bias, logits, T, U1, V, ldl, slope);
        else
            joint_fwd_regw_kernel<40, false><<<dim3(nut, tsplit, B), 256, lds_r, st>>>((const bf16_t *)enc, (const bf16_t *)dec, W, bias, logits, T, U1, V, ldl, slope);
    } else if (io_dtype == TSASR_BF16) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)joint_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        joint_fwd_kernel<bf16_t><<<grid, 256, lds, st>>>((const bf16_t *)enc, (const bf16_t *)dec, W, bias, logits, T, U1, J, V, ldl, slope);
    } else {
        TSASR_CHECK_ARG(false, "tsasr_joint_fwd: bad io_dtype %d", io_dtype);
    }
    TSASR_CHECK_LAUNCH("tsasr_joint_fwd");
    return 0;
}

}  // extern "C"

// denc[e] = sum over the u tiles of part[tile][e] (fixed order), four elements per thread, written in the io dtype
template <typename T>
__global__ __launch_bounds__(256) void denc_sum_kernel(const float *__restrict__ part, T *__restrict__ denc, long long n, int nparts) {
    for (long long e = (blockIdx.x * 256LL + threadIdx.x) * 4; e < n; e += (long long)gridDim.x * 256 * 4) {
        float4 s = *reinterpret_cast<const float4 *>(part + e);
        for (int p = 1; p < nparts; ++p) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (long long)p * n + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        st1(denc + e, s.x); st1(denc + e + 1, s.y); st1(denc + e + 2, s.z); st1(denc + e + 3, s.w);
    }
}

extern "C" {

static int device_cu_count_rnnt() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// frame ranges per (utterance, u tile, k group). Measured on the benchmark step (B = 32, T' = 250, 256 workgroups = one per CU without
// a split): 1 -> 0.66 ms, 2 -> 0.75, 3 -> 0.84, 4 -> 0.93 - a second resident workgroup per CU buys nothing (the kernel is bound by
// what a CU issues per frame, not by latency) and every range repeats the fragment loads and the slab writes. So: split only when the
// grid would leave CUs idle (B = 1 long-form: 120 workgroups), never below 32 frames per range; TSASR_JOINT_TSPLIT forces a value.
static int joint_tsplit(int B, int T, int U1, int J) {
    static const int forced = [] { const char *e = getenv("TSASR_JOINT_TSPLIT"); return e ? atoi(e) : 0; }();
    if (forced > 0) return std::min(forced, 4);
    const long long wgs = (long long)B * cdiv(U1, 32) * cdiv(J / 32, 4 * KB);
    int ts = (int)std::min<long long>(4, (long long)JBX_OCC * device_cu_count_rnnt() / std::max<long long>(wgs, 1));
    while (ts > 1 && T / ts < 32) --ts;
    return std::max(ts, 1);
}

size_t tsasr_joint_bwd_workspace_bytes(int B, int T, int U1, int J) {
    const size_t nslab = (size_t)B * cdiv(U1, 32) * 4;        // sized for the largest frame split
    return align_up(nslab * 32 * J * sizeof(float), 256) + align_up(nslab * 32 * sizeof(float), 256) +
           align_up((size_t)B * cdiv(U1, 32) * T * J * sizeof(float), 256) +      // denc partial sums per u tile
           align_up((size_t)4 * B * U1 * J * sizeof(float), 256);                 // ddec partial sums per frame range
}

int tsasr_joint_bwd(const float *dlogits, const void *enc, const void *dec, const float *W, void *denc, void *ddec,
                    float *dW, float *dbias, const int32_t *tlen, const int32_t *ulen, int B, int T, int U1, int J, int V,
                    int ldl, int io_dtype, float slope, void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(dlogits && enc && dec && W && denc && ddec && dW && dbias && workspace, "tsasr_joint_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0, "tsasr_joint_bwd: empty batch");
    TSASR_CHECK_ARG(J > 0 && J % 32 == 0, "tsasr_joint_bwd: J=%d must be a multiple of 32", J);
    TSASR_CHECK_ARG(V > 0 && V <= 32 && ldl >= V && ldl % 4 == 0 && ldl <= 32, "tsasr_joint_bwd: V=%d ldl=%d unsupported", V, ldl);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_joint_bwd_workspace_bytes(B, T, U1, J), "tsasr_joint_bwd: workspace too small");
    const int nut = cdiv(U1, 32), nkb = J / 32;
    const int ksplit = cdiv(nkb, 4 * KB);
    const int TS = joint_tsplit(B, T, U1, J);
    const size_t nslab = (size_t)B * nut * TS, nslab_max = (size_t)B * nut * 4;
    float *slab_w = (float *)workspace;
    float *slab_b = (float *)((char *)workspace + align_up(nslab_max * 32 * J * sizeof(float), 256));
    hipStream_t st = (hipStream_t)stream;
    dim3 gx(nut, ksplit, B * TS), gy(cdiv(T, TG), ksplit, B);
    // one pass: the kernel that forms dh tile by tile for ddec / dW also leaves each tile's column sums (its share of denc); a small
    // kernel adds the cdiv(U1, 32) shares. TSASR_JOINT_TWO_PASS=1 brings the separate denc kernel back (A/B: it recomputes every tile).
    static const bool two_pass = false;
    char *after_slabs = (char *)workspace + align_up(nslab_max * 32 * J * sizeof(float), 256) + align_up(nslab_max * 32 * sizeof(float), 256);
    float *part = two_pass ? nullptr : (float *)after_slabs;
    float *ddec_part = (float *)(after_slabs + align_up((size_t)B * nut * T * J * sizeof(float), 256));
    const long long n_enc = (long long)B * T * J, n_dec = (long long)B * U1 * J;
    if (io_dtype == TSASR_F32) {
        joint_bwd_x_kernel<float><<<gx, 256, 0, st>>>(dlogits, (const float *)enc, (const float *)dec, W, (float *)ddec, slab_w, slab_b, tlen, ulen, T, U1, J, V, ldl, slope, part, TS, ddec_part);
        if (TS > 1) denc_sum_kernel<float><<<(unsigned)std::min<long long>(4096, (n_dec / 4 + 255) / 256), 256, 0, st>>>(ddec_part, (float *)ddec, n_dec, TS);
        if (two_pass) joint_bwd_y_kernel<float><<<gy, 256, 0, st>>>(dlogits, (const float *)enc, (const float *)dec, W, (float *)denc, tlen, ulen, T, U1, J, V, ldl, slope);
        else denc_sum_kernel<float><<<(unsigned)std::min<long long>(4096, (n_enc / 4 + 255) / 256), 256, 0, st>>>(part, (float *)denc, n_enc, nut);
    } else if (io_dtype == TSASR_BF16) {
        joint_bwd_x_kernel<bf16_t><<<gx, 256, 0, st>>>(dlogits, (const bf16_t *)enc, (const bf16_t *)dec, W, (bf16_t *)ddec, slab_w, slab_b, tlen, ulen, T, U1, J, V, ldl, slope, part, TS, ddec_part);
        if (TS > 1) denc_sum_kernel<bf16_t><<<(unsigned)std::min<long long>(4096, (n_dec / 4 + 255) / 256), 256, 0, st>>>(ddec_part, (bf16_t *)ddec, n_dec, TS);
        if (two_pass) joint_bwd_y_kernel<bf16_t><<<gy, 256, 0, st>>>(dlogits, (const bf16_t *)enc, (const bf16_t *)dec, W, (bf16_t *)denc, tlen, ulen, T, U1, J, V, ldl, slope);
        else denc_sum_kernel<bf16_t><<<(unsigned)std::min<long long>(4096, (n_enc / 4 + 255) / 256), 256, 0, st>>>(part, (bf16_t *)denc, n_enc, nut);
    } else {
        TSASR_CHECK_ARG(false, "tsasr_joint_bwd: bad io_dtype %d", io_dtype);
    }
    static const bool joint_defer = true;
    if (tsasr_reduce_deferring() && joint_defer) {   // head-weight slabs join the batched reduction at the end of backward (the loop kernel below walks
        tsasr_reduce_submit(slab_w, dW, (long long)32 * J, (int)nslab, V * J, 0, st);   // 128 slabs serially per thread: 57 us on the main stream)
        tsasr_reduce_submit(slab_b, dbias, 32, (int)nslab, V, 0, st);
    } else {
        joint_bwd_reduce_kernel<<<cdiv(V * J, 256), 256, 0, st>>>(slab_w, slab_b, dW, dbias, (int)nslab, J, V);
    }
    TSASR_CHECK_LAUNCH("tsasr_joint_bwd");
    return 0;
}

/* Lab / test switch: how the next tsasr_rnnt_loss_* calls lay out and walk the lattice (see rnnt_plan; -1 = default for each):
 * skew 0 / 1 / 2 = frame-major planes always / skewed when a lattice runs on more than one wave / always; waves = cap on the waves of a
 * one-workgroup lattice; mc = 0 never split a lattice over workgroups, 1 / 2 / 4 = split whenever it fits, that many columns per thread.
 * Changes tsasr_rnnt_loss_workspace_bytes; forward and backward of one loss must run under the same plan. Every plan gives the same bits. */
void tsasr_rnnt_lattice_plan(int skew, int waves, int mc) { g_plan_skew = skew; g_plan_waves = waves; g_plan_mc = mc; }

/* Byte offset, inside a loss workspace, of the split lattice's error word (uint32: 1 = a workgroup's wait for its neighbour ran out and the
 * costs of that launch are NaN; any other value = fine), or -1 when this shape does not split its lattice over workgroups. */
long long tsasr_rnnt_loss_error_word_offset(int B, int T, int U1) {
    if (B <= 0 || T <= 0 || U1 <= 0) return -1;
    char base[1];
    const RnntWs w = rnnt_carve(base, B, T, U1);
    return w.err ? (long long)(reinterpret_cast<char *>(w.err) - base) : -1;
}

size_t tsasr_rnnt_loss_workspace_bytes(int B, int T, int U1) {
    if (B <= 0 || T <= 0 || U1 <= 0) return 0;
    return align_up((6 * rnnt_plane_floats(B, T, U1) + align_up((size_t)B, 64) + rnnt_xch_floats(B, T, U1)) * sizeof(float), 256);
}

int tsasr_rnnt_loss_fwd(const float *logits, const int32_t *targets, int ldt, const int32_t *tlen, const int32_t *ulen,
                        float *costs, int B, int T, int U1, int V, int ldl, int blank, void *workspace,
                        size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(logits && targets && tlen && ulen && costs && workspace, "tsasr_rnnt_loss_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0, "tsasr_rnnt_loss_fwd: empty batch (B=%d T=%d U1=%d)", B, T, U1);
    TSASR_CHECK_ARG(V > 0 && ldl >= V && ldl % 4 == 0, "tsasr_rnnt_loss_fwd: rows must be padded to a multiple of 4 floats (V=%d ldl=%d)", V, ldl);
    TSASR_CHECK_ARG(blank >= 0 && blank < V, "tsasr_rnnt_loss_fwd: blank=%d outside [0,%d)", blank, V);
    TSASR_CHECK_ARG(U1 <= 64 * 32, "tsasr_rnnt_loss_fwd: U1=%d > 2048 lattice columns not supported", U1);
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_rnnt_loss_workspace_bytes(B, T, U1), "tsasr_rnnt_loss_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    RnntWs w = rnnt_carve(workspace, B, T, U1);
    const long long rows = (long long)B * T * U1;
    rnnt_lp_kernel<<<(unsigned)(V <= 32 ? (rows + 127) / 128 : (rows + 31) / 32), 256, 0, st>>>(logits, targets, ldt, tlen, ulen, w, B, T, U1, V, ldl, blank);
    if (int rc = launch_alphabeta(w, tlen, ulen, costs, B, T, U1, st)) return rc;
    TSASR_CHECK_LAUNCH("tsasr_rnnt_loss_fwd");
    return 0;
}

int tsasr_rnnt_loss_bwd(const float *logits, const int32_t *targets, int ldt, const int32_t *tlen, const int32_t *ulen,
                        const float *gscale, float *dlogits, int B, int T, int U1, int V, int ldl, int blank,
                        const void *workspace, size_t workspace_bytes, void *stream) {
    TSASR_CHECK_ARG(logits && targets && tlen && ulen && gscale && dlogits && workspace, "tsasr_rnnt_loss_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && V > 0 && ldl >= V && ldl % 4 == 0, "tsasr_rnnt_loss_bwd: bad shape");
    TSASR_CHECK_ARG(workspace_bytes >= tsasr_rnnt_loss_workspace_bytes(B, T, U1), "tsasr_rnnt_loss_bwd: workspace too small");
    RnntWs w = rnnt_carve(const_cast<void *>(workspace), B, T, U1);
    const long long rows = (long long)B * T * U1;
    rnnt_grad_kernel<<<(unsigned)((rows + 31) / 32), 256, 0, (hipStream_t)stream>>>(logits, targets, ldt, tlen, ulen, gscale, dlogits, w, B, T, U1, V, ldl, blank);
    TSASR_CHECK_LAUNCH("tsasr_rnnt_loss_bwd");
    return 0;
}

}  // extern "C"
