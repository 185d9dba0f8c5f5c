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

}  // extern "C"
