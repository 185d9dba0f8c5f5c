// LSTM cell kernels for the transducer's prediction network on gfx950 (one launch per time step and direction; the
// recurrent matrix product runs on csrc/gemm.hip - every launch is hipGraph-capturable, unlike MIOpen's RNN path).
//
// Replaces torch.nn.LSTM as wrapped by speechbrain/nnet/RNN.py:244-278 (single layer, batch_first, gate order i,f,g,o).
//   gates_t = x_t W_ih^T + b_ih + b_hh + h_{t-1} W_hh^T ;  c_t = sig(f) c_{t-1} + sig(i) tanh(g) ;  h_t = sig(o) tanh(c_t)
// Layout: gates [B, U, 4H] fp32 (pre-activations in, ACTIVATED gates out - kept for the backward), c [B, U, H] fp32,
// h [B, U, H] in the activation dtype (it is the next step's GEMM operand and the layer output).
#include "common.h"

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { const float e = __expf(-2.f * fabsf(x)); const float t = (1.f - e) / (1.f + e); return x < 0.f ? -t : t; }

template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float *__restrict__ gates, float *__restrict__ c, T *__restrict__ h, int B,
                                                            int U, int H, int t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    float *g = gates + ((long long)b * U + t) * 4 * H;
    const float gi = sigm(g[k]), gf = sigm(g[H + k]), gg = tanh_fast(g[2 * H + k]), go = sigm(g[3 * H + k]);
    const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
    const float cn = gf * cp + gi * gg;
    g[k] = gi; g[H + k] = gf; g[2 * H + k] = gg; g[3 * H + k] = go;
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
    const float *g = gates + ((long long)b * U + t) * 4 * H;
    const float gi = g[k], gf = g[H + k], gg = g[2 * H + k], go = g[3 * H + k];
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
        float *gp = gates + ((long long)b * U + t) * 4 * H;
        const float gi = sigm(gp[k] + pre[0]), gf = sigm(gp[H + k] + pre[1]), gg = tanh_fast(gp[2 * H + k] + pre[2]), go = sigm(gp[3 * H + k] + pre[3]);
        const float cp = t > 0 ? c[((long long)b * U + t - 1) * H + k] : 0.f;
        const float cn = gf * cp + gi * gg;
        gp[k] = gi; gp[H + k] = gf; gp[2 * H + k] = gg; gp[3 * H + k] = go;
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
        const float *g = gates + ((long long)b * U + t) * 4 * H;
        const float gi = g[k], gf = g[H + k], gg = g[2 * H + k], go = g[3 * H + k];
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

}  // extern "C"
