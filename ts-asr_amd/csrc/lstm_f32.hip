// Single-layer unidirectional LSTM in EXACT fp32 arithmetic, sequence form with optional initial state: the `compute_dtype: fp32` parity
// mode of the predictor and the step-wise predictor calls of the transducer searchers (one token at a time with a carried (h, c)).
//
// Replaces torch.nn.LSTM as SB/nnet/RNN.py:170-278 wraps it (training, fp32) and as SB/decoders/transducer.py:246-353,411-466 steps it
// during greedy / beam search (`_forward_PN`: decode_network_lst[0..](input, hidden)); until round 4 both went through MIOpen. The bf16
// training path is csrc/lstm.hip (persistent MFMA kernels); this file is the checker-grade fp32 path: no matrix cores, no rounding.
// Gate order i, f, g, o and the two bias vectors as torch.nn.LSTM.
//   forward : one workgroup (1024 threads) per batch row walks the U steps: wave w computes the gate rows w, w + 16, ... as dot products
//             over [x_t | h_{t-1}] (lanes stride the inner index: coalesced 256-byte reads of the weight rows), then thread j < H forms
//             (i, f, g, o), c_t, h_t; c stays in a register, h goes through LDS. Saves the ACTIVATED gates and c for the backward.
//   backward: one workgroup per batch row walks t = U-1 .. 0: gate pre-activation gradients from the saved gates, then
//             dh_{t-1}[k] = sum_r W_hh[r][k] dpre[r] (thread k, rows from LDS broadcasts). dgates leave as [B,U,4H]; the weight / bias /
//             input gradients are three fp32 GEMMs over all (b, t) on the host side (csrc/gemm_f32.hip), as in the bf16 path.
#include "common.h"

namespace {

constexpr int LT = 1024;      // threads per workgroup

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(LT) void lstm_f32_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                                          const float *__restrict__ b_ih, const float *__restrict__ b_hh,
                                                          const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ hs,
                                                          float *__restrict__ cs, float *__restrict__ gates, float *__restrict__ hn,
                                                          float *__restrict__ cn, int U, int I, int H) {
    extern __shared__ float sm[];
    float *xh = sm;               // [I + H]: x_t then h_{t-1}
    float *pre = sm + I + H;      // [4H]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = 4 * H, KK = I + H;
    float c = 0.f;
    if (tid < H) {
        xh[I + tid] = h0 ? h0[(size_t)b * H + tid] : 0.f;
        c = c0 ? c0[(size_t)b * H + tid] : 0.f;
    }
    for (int t = 0; t < U; ++t) {
        for (int k = tid; k < I; k += LT) xh[k] = x[((size_t)b * U + t) * I + k];
        __syncthreads();
        for (int r = wave; r < G; r += LT / 64) {
            float s = 0.f;
            const float *wi = w_ih + (size_t)r * I, *wh = w_hh + (size_t)r * H;
            for (int k = lane; k < I; k += 64) s += wi[k] * xh[k];
            for (int k = lane; k < H; k += 64) s += wh[k] * xh[I + k];
            s = wave_sum(s);
            if (lane == 0) pre[r] = s + (b_ih ? b_ih[r] : 0.f) + (b_hh ? b_hh[r] : 0.f);
        }
        __syncthreads();
        if (tid < H) {
            const float gi = sigm(pre[tid]), gf = sigm(pre[H + tid]), gg = tanhf(pre[2 * H + tid]), go = sigm(pre[3 * H + tid]);
            c = gf * c + gi * gg;
            const float h = go * tanhf(c);
            const size_t o = ((size_t)b * U + t) * H + tid;
            hs[o] = h;
            if (cs) cs[o] = c;
            if (gates) {
                float *gp = gates + ((size_t)b * U + t) * G;
                gp[tid] = gi; gp[H + tid] = gf; gp[2 * H + tid] = gg; gp[3 * H + tid] = go;
            }
            xh[I + tid] = h;
            if (t == U - 1) {
                if (hn) hn[(size_t)b * H + tid] = h;
                if (cn) cn[(size_t)b * H + tid] = c;
            }
        }
        __syncthreads();
        (void)KK;
    }
}

__global__ __launch_bounds__(LT) void lstm_f32_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ dhn, const float *__restrict__ dcn,
                                                          const float *__restrict__ gates, const float *__restrict__ cs, const float *__restrict__ c0,
                                                          const float *__restrict__ w_hh, float *__restrict__ dgates, float *__restrict__ dh0,
                                                          float *__restrict__ dc0, int U, int H) {
    extern __shared__ float sm[];
    float *dpre = sm;             // [4H]
    float *dhl = sm + 4 * H;      // [H] gradient flowing into h_t from step t + 1
    const int b = blockIdx.x, tid = threadIdx.x;
    const int G = 4 * H;
    float dc = 0.f;
    if (tid < H) {
        dhl[tid] = dhn ? dhn[(size_t)b * H + tid] : 0.f;
        dc = dcn ? dcn[(size_t)b * H + tid] : 0.f;
    }
    __syncthreads();
    for (int t = U - 1; t >= 0; --t) {
        if (tid < H) {
            const size_t o = ((size_t)b * U + t) * H + tid;
            const float *gp = gates + ((size_t)b * U + t) * G;
            const float gi = gp[tid], gf = gp[H + tid], gg = gp[2 * H + tid], go = gp[3 * H + tid];
            const float ct = cs[o], cp = t > 0 ? cs[o - H] : (c0 ? c0[(size_t)b * H + tid] : 0.f);
            const float dh = (dout ? dout[o] : 0.f) + dhl[tid];
            const float tc = tanhf(ct);
            dc += dh * go * (1.f - tc * tc);
            const float di = dc * gg * gi * (1.f - gi), df = dc * cp * gf * (1.f - gf), dg = dc * gi * (1.f - gg * gg), dO = dh * tc * go * (1.f - go);
            dc = dc * gf;
            dpre[tid] = di; dpre[H + tid] = df; dpre[2 * H + tid] = dg; dpre[3 * H + tid] = dO;
            float *dgp = dgates + ((size_t)b * U + t) * G;
            dgp[tid] = di; dgp[H + tid] = df; dgp[2 * H + tid] = dg; dgp[3 * H + tid] = dO;
        }
        __syncthreads();
        float acc = 0.f;
        if (tid < H)
            for (int r = 0; r < G; ++r) acc += w_hh[(size_t)r * H + tid] * dpre[r];
        __syncthreads();
        if (tid < H) dhl[tid] = acc;
        __syncthreads();
    }
    if (tid < H) {
        if (dh0) dh0[(size_t)b * H + tid] = dhl[tid];
        if (dc0) dc0[(size_t)b * H + tid] = dc;
    }
}

}  // namespace

extern "C" {

/* hs [B,U,H] (+ hn, cn [B,H]: the state after the last step; cs [B,U,H] and gates [B,U,4H] = activated (i,f,g,o): saved for the backward,
 * NULL when not training) of a single-layer LSTM over x [B,U,I] from the state (h0, c0) (NULL = zeros). H <= 1024. */
int tsasr_lstm_f32_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *h0,
                       const float *c0, float *hs, float *cs, float *gates, float *hn, float *cn, int B, int U, int I, int H, void *stream) {
    TSASR_CHECK_ARG(x && w_ih && w_hh && hs, "tsasr_lstm_f32_fwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && U > 0 && I > 0 && H > 0 && H <= LT, "tsasr_lstm_f32_fwd: bad sizes B=%d U=%d I=%d H=%d (H <= %d)", B, U, I, H, LT);
    const size_t lds = (size_t)(I + H + 4 * H) * sizeof(float);
    TSASR_CHECK_ARG(lds <= 160 * 1024, "tsasr_lstm_f32_fwd: I=%d H=%d need %zu bytes of LDS", I, H, lds);
    (void)hipFuncSetAttribute((const void *)lstm_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lstm_f32_fwd_kernel<<<B, LT, lds, (hipStream_t)stream>>>(x, w_ih, w_hh, b_ih, b_hh, h0, c0, hs, cs, gates, hn, cn, U, I, H);
    TSASR_CHECK_LAUNCH("tsasr_lstm_f32_fwd");
    return 0;
}

/* dgates [B,U,4H] = gradients w.r.t. the gate PRE-activations (dW_ih = dgates^T x, dW_hh = dgates^T h_prev, db = column sums, dx = dgates W_ih),
 * dh0 / dc0 [B,H] (may be NULL) from dout [B,U,H] (may be NULL) and the gradients dhn / dcn [B,H] of the final state (may be NULL). */
int tsasr_lstm_f32_bwd(const float *dout, const float *dhn, const float *dcn, const float *gates, const float *cs, const float *c0,
                       const float *w_hh, float *dgates, float *dh0, float *dc0, int B, int U, int H, void *stream) {
    TSASR_CHECK_ARG(gates && cs && w_hh && dgates, "tsasr_lstm_f32_bwd: null pointer");
    TSASR_CHECK_ARG(B > 0 && U > 0 && H > 0 && H <= LT, "tsasr_lstm_f32_bwd: bad sizes");
    const size_t lds = (size_t)(5 * H) * sizeof(float);
    lstm_f32_bwd_kernel<<<B, LT, lds, (hipStream_t)stream>>>(dout, dhn, dcn, gates, cs, c0, w_hh, dgates, dh0, dc0, U, H);
    TSASR_CHECK_LAUNCH("tsasr_lstm_f32_bwd");
    return 0;
}

}  // extern "C"
