// Greedy transducer search on the device (gfx950): one persistent workgroup per utterance walks the encoder frames.
//
// Replaces speechbrain/decoders/transducer.py:138-218 (transducer_greedy_decode): per frame the joint of the frame with the current
// predictor output, the classifier, log-softmax, argmax; an utterance whose best symbol is not blank appends it and advances its
// predictor (embedding -> one LSTM step -> projection) - at most one symbol per frame. The reference (and ts-asr_amd/decoders.py's host
// loop, kept for other network shapes) issues ~15 library launches per frame; here the predictor state (h, c, projected output)
// stays in LDS for the whole utterance and one launch decodes the batch.
//
//   logits[v] = b_head[v] + sum_k W_head[v][k] * LeakyReLU(enc[b,t,k] + pn[k])            (Transducer_joint "sum" + Linear head)
//   LSTM step (torch gate order i, f, g, o):  gates = W_ih x + b_ih + W_hh h + b_hh,  x = emb[token]
//   pn = W_proj h + b_proj
// Matrix-vector products: one wave per output row (rows w, w+4, ...), lanes split the inner dimension in 16-byte pieces (whole rows
// are read as contiguous lines, from L2: the 2-4 MB of recurrent weights are shared by all utterances), DPP wave reduction, the row's
// value lands in LDS. All arithmetic fp32; weights fp32 (master parameters) or bf16 (the training step's shadow copies).
#include "common.h"

namespace {

constexpr int GREEDY_THREADS = 1024;   // 16 waves: the matrix-vector products are latency chains per wave

struct GreedyArgs {
    const void *enc;            // [B, T, J] io dtype
    const float *emb;           // [V_emb, E] fp32 (one-hot table or learned)
    const void *w_ih, *w_hh;    // [4H, E], [4H, H]  (wdtype)
    const float *b_ih, *b_hh;   // [4H] fp32 (may be NULL)
    const void *w_proj;         // [J, H] (wdtype)
    const float *b_proj;        // [J] or NULL
    const void *w_head;         // [V, J] (wdtype)
    const float *b_head;        // [V] or NULL
    int *preds;                 // [B, T]: symbol emitted at frame t, -1 = blank
    float *logp_sum;            // [B]: sum of the emitted symbols' log-probabilities
    int B, T, J, H, E, V, blank;
    float slope;
};

template <typename WT> __device__ __forceinline__ void ld4w(const WT *p, float (&o)[4]);
template <> __device__ __forceinline__ void ld4w<float>(const float *p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ld4w<bf16_t>(const bf16_t *p, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// out[r] = bias[r] (+ bias2[r]) + sum_k W[r][k] x[k]  (+ sum_e W2[r][e] x2[e], E2 <= 64), r < rows; x in LDS, K % 4 == 0, K <= 1024.
// A wave takes RB rows per pass: all their 16-byte pieces are requested before the first is used (one memory round trip per RB rows,
// not per row: with one row at a time a predictor step took 380 us), the lane's slice of x sits in registers for the whole call.
template <typename WT, int RB, int NC>                // NC = pieces of 256 columns per row (K <= 256 * NC)
__device__ __forceinline__ void gemv_rows_nc(const WT *__restrict__ W, int rows, int K, const float *x, const float *__restrict__ bias,
                                             const float *__restrict__ bias2, const WT *__restrict__ W2, int E2, const float *x2, float *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const float xe = (W2 && lane < E2) ? x2[lane] : 0.f;
    float xr[NC][4];                                  // x[k], k = c * 256 + lane * 4 .. +3
    constexpr int nc = NC;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = c * 256 + lane * 4 + q;
            xr[c][q] = k < K ? x[k] : 0.f;
        }
    for (int r0 = wave * RB; r0 < rows; r0 += nw * RB) {
        float w[RB][NC][4], w2[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = min(r0 + i, rows - 1);      // clamped: always issued, the surplus rows are not stored
            const WT *wr = W + (size_t)r * K;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = min(c * 256 + lane * 4, K - 4);
                ld4w<WT>(wr + k, w[i][c]);
            }
            w2[i] = W2 ? (float)W2[(size_t)r * E2 + min(lane, E2 - 1)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float acc = (W2 && lane < E2) ? w2[i] * xe : 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc += w[i][c][0] * xr[c][0] + w[i][c][1] * xr[c][1] + w[i][c][2] * xr[c][2] + w[i][c][3] * xr[c][3];
            acc = wave_sum(acc);
            const int r = r0 + i;
            if (lane == 0 && r < rows) out[r] = acc + (bias ? bias[r] : 0.f) + (bias2 ? bias2[r] : 0.f);
        }
    }
}

template <typename WT>
__device__ __forceinline__ void gemv_rows(const WT *__restrict__ W, int rows, int K, const float *x, const float *__restrict__ bias,
                                          const float *__restrict__ bias2, const WT *__restrict__ W2, int E2, const float *x2, float *out) {
    if (K <= 512) gemv_rows_nc<WT, 8, 2>(W, rows, K, x, bias, bias2, W2, E2, x2, out);         // (registers: RB * NC * 4 weights in flight)
    else if (K <= 768) gemv_rows_nc<WT, 4, 3>(W, rows, K, x, bias, bias2, W2, E2, x2, out);
    else gemv_rows_nc<WT, 4, 4>(W, rows, K, x, bias, bias2, W2, E2, x2, out);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T, typename WT>
__global__ __launch_bounds__(GREEDY_THREADS) void greedy_decode_kernel(const GreedyArgs A) {
    extern __shared__ __attribute__((aligned(16))) float gs[];
    float *h = gs, *c = h + A.H, *pn = c + A.H, *z = pn + A.J, *gates = z + A.J, *x = gates + 4 * A.H, *logits = x + 64;
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x;
    const T *enc = (const T *)A.enc + (size_t)b * A.T * A.J;
    const WT *w_ih = (const WT *)A.w_ih, *w_hh = (const WT *)A.w_hh, *w_proj = (const WT *)A.w_proj, *w_head = (const WT *)A.w_head;
    for (int i = tid; i < A.H; i += GREEDY_THREADS) h[i] = c[i] = 0.f;
    if (tid == 0) s_tok = A.blank;
    float lsum = 0.f;
    __syncthreads();
    // step(-1): the predictor is primed with the blank symbol (transducer.py:160-170); then one pass per frame
    for (int t = -1; t < A.T; ++t) {
        int tok = s_tok;                        // symbol to feed the predictor with (every thread reads the same LDS word)
        bool advance = (t < 0);
        if (t >= 0) {
            for (int k = tid; k < A.J; k += GREEDY_THREADS) {
                const float v = ld1(enc + (size_t)t * A.J + k) + pn[k];
                z[k] = v > 0.f ? v : v * A.slope;
            }
            __syncthreads();
            gemv_rows<WT>(w_head, A.V, A.J, z, A.b_head, nullptr, nullptr, 0, nullptr, logits);
            __syncthreads();
            if (tid < 64) {                     // arg max (lowest index among ties) and its log-probability, one wave
                const float v = tid < A.V ? logits[tid] : -INFINITY;
                float m = v;
                m = wave_max(m);
                const unsigned long long eq = __ballot(v == m && tid < A.V);
                const int pos = __ffsll((long long)eq) - 1;
                const float s = wave_sum(tid < A.V ? __expf(v - m) : 0.f);
                if (tid == 0) {
                    const bool emit = pos != A.blank;
                    A.preds[(size_t)b * A.T + t] = emit ? pos : -1;
                    if (emit) { lsum += -__logf(s); s_tok = pos; }     // log-softmax at the maximum = -log(sum exp(v - m))
                    logits[63] = emit ? 1.f : 0.f;
                }
            }
            __syncthreads();
            advance = logits[63] != 0.f;
            tok = s_tok;
        }
        if (advance) {                          // workgroup-uniform
            for (int e = tid; e < A.E; e += GREEDY_THREADS) x[e] = A.emb[(size_t)tok * A.E + e];
            __syncthreads();
            gemv_rows<WT>(w_hh, 4 * A.H, A.H, h, A.b_ih, A.b_hh, w_ih, A.E, x, gates);
            __syncthreads();
            for (int u = tid; u < A.H; u += GREEDY_THREADS) {
                const float ig = sigmoid_f(gates[u]), fg = sigmoid_f(gates[A.H + u]), gg = tanhf(gates[2 * A.H + u]), og = sigmoid_f(gates[3 * A.H + u]);
                const float cn = fg * c[u] + ig * gg;
                c[u] = cn;
                h[u] = og * tanhf(cn);
            }
            __syncthreads();
            gemv_rows<WT>(w_proj, A.J, A.H, h, A.b_proj, nullptr, nullptr, 0, nullptr, pn);
            __syncthreads();
        }
    }
    if (tid == 0) A.logp_sum[b] = lsum;
}

}  // namespace

extern "C" {

/* Greedy transducer search (speechbrain/decoders/transducer.py:138-218) for a predictor = embedding table -> one-layer LSTM -> Linear and
 * a joiner = LeakyReLU(enc + pn) -> Linear head. enc [B,T,J] (io_dtype); emb fp32 [n_emb, E] (E <= 64); LSTM weights in torch layout
 * (w_ih [4H,E], w_hh [4H,H], gate order i,f,g,o); w_proj [J,H]; w_head [V,J] (V <= 64); matrices in `wdtype` (TSASR_F32 / TSASR_BF16),
 * biases fp32 or NULL. H, J multiples of 4 and <= 1024. preds int32 [B,T]: the symbol emitted at each frame or -1; logp_sum fp32 [B]. One launch. */
int tsasr_greedy_decode(const void *enc, const float *emb, const void *w_ih, const void *w_hh, const float *b_ih, const float *b_hh,
                        const void *w_proj, const float *b_proj, const void *w_head, const float *b_head, int *preds, float *logp_sum,
                        int B, int T, int J, int H, int E, int V, int blank, float slope, int io_dtype, int wdtype, void *stream) {
    TSASR_CHECK_ARG(enc && emb && w_ih && w_hh && w_proj && w_head && preds && logp_sum, "tsasr_greedy_decode: null pointer");
    TSASR_CHECK_ARG(B > 0 && T > 0 && J > 0 && J % 4 == 0 && H > 0 && H % 4 == 0 && E > 0 && E <= 64 && V > 1 && V <= 63 && blank >= 0 && blank < V,
                    "tsasr_greedy_decode: bad shape (B=%d T=%d J=%d H=%d E=%d V=%d blank=%d)", B, T, J, H, E, V, blank);
    TSASR_CHECK_ARG(J <= 1024 && H <= 1024, "tsasr_greedy_decode: J=%d H=%d above 1024", J, H);
    TSASR_CHECK_ARG((io_dtype == TSASR_F32 || io_dtype == TSASR_BF16) && (wdtype == TSASR_F32 || wdtype == TSASR_BF16), "tsasr_greedy_decode: bad dtype");
    const size_t lds = (size_t)(2 * H + 2 * J + 4 * H + 64 + 64) * sizeof(float);
    TSASR_CHECK_ARG(lds <= 160 * 1024, "tsasr_greedy_decode: H=%d J=%d need %zu B of LDS", H, J, lds);
    GreedyArgs a{enc, emb, w_ih, w_hh, b_ih, b_hh, w_proj, b_proj, w_head, b_head, preds, logp_sum, B, T, J, H, E, V, blank, slope};
    hipStream_t st = (hipStream_t)stream;
#define GREEDY(TT, WW)                                                                                                          \
    {                                                                                                                           \
        auto kern = greedy_decode_kernel<TT, WW>;                                                                               \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        kern<<<B, GREEDY_THREADS, lds, st>>>(a);                                                                                           \
    }
    if (io_dtype == TSASR_F32) { if (wdtype == TSASR_F32) GREEDY(float, float) else GREEDY(float, bf16_t) }
    else { if (wdtype == TSASR_F32) GREEDY(bf16_t, float) else GREEDY(bf16_t, bf16_t) }
#undef GREEDY
    TSASR_CHECK_LAUNCH("tsasr_greedy_decode");
    return 0;
}

}  // extern "C"
