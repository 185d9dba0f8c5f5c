/* libtsasr_hip.so - C-ABI of the MI355X (gfx950) TS-ASR Conformer-Transducer hot path.
 *
 * The reference (lucadellalib/ts-asr) is 100 % Python: it has no FFI for this path. The boundary a
 * maintainer binds is therefore the set of torch.nn.Module / loss callables the hparams YAML
 * instantiates; every entry point below names the reference call it
 * replaces (paths relative to the reference root; SB = vendor/speechbrain/speechbrain).
 * The ctypes stubs a maintainer would add are in INTEGRATION.md; ts-asr_amd/_capi.py holds ours.
 *
 * Conventions
 *   - plain pointers and sizes; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t; nothing allocates, nothing synchronises, nothing touches the
 *     default stream: calls are graph-capturable and re-entrant per stream;
 *   - scratch memory comes from the caller: `*_workspace_bytes()` says how much;
 *   - return value: 0 = ok, <0 = error (TSASR_E_*); tsasr_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - io_dtype: TSASR_F32 (0) or TSASR_BF16 (1) = storage type of activations; contractions use
 *     bf16 MFMA operands with fp32 accumulation, every reduction / softmax / LSE is fp32;
 *   - lengths are ABSOLUTE int32 counts on the device (the host mirror converts the reference's
 *     relative lengths with the reference's own rounding rule before the call).
 */
#ifndef TSASR_HIP_H
#define TSASR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSASR_F32 0
#define TSASR_BF16 1

#define TSASR_E_INVALID (-1)  /* bad shape / unsupported size / null pointer */
#define TSASR_E_LAUNCH (-2)   /* hipLaunch / runtime failure */
#define TSASR_E_WORKSPACE (-3)

const char *tsasr_last_error(void);
int tsasr_version(void);
/* 1 when a gfx950 device is current, 0 otherwise (no kernel is launched). */
int tsasr_device_ok(void);

/* ------------------------------------------------------------------------------------------
 * RNN-T joint + head  (replaces SB/nnet/transducer/transducer_joint.py:73-95 `Transducer_joint.forward`
 * (joint="sum", LeakyReLU) followed by SB/nnet/linear.py:64-78 `Linear.forward` of `transducer_head`,
 * call site train_librispeechmix_scratch.py:132-135) - the [B,T,U1,J] joint tensor is never materialised.
 *   logits[b,t,u,v] = bias[v] + sum_k W[v,k] * lrelu(enc[b,t,k] + dec[b,u,k])      v < V <= 32
 * logits rows are padded to `ldl` floats (ldl % 4 == 0, ldl >= V; columns >= V are written as 0).
 * ------------------------------------------------------------------------------------------ */
int tsasr_joint_fwd(const void *enc, const void *dec, const float *W, const float *bias, float *logits,
                    int B, int T, int U1, int J, int V, int ldl, int io_dtype, float slope, void *stream);

/* The same joint + head in EXACT fp32 arithmetic (csrc/joint_f32.hip: no matrix cores, no bf16 rounding of h, W or dlogits): the
 * `compute_dtype: fp32` parity mode. fp32 storage only; V <= 32, J <= 768; dW / dbias through the deferrable batched reduction. */
int tsasr_joint_f32_fwd(const float *enc, const float *dec, const float *W, const float *bias, float *logits, int B, int T, int U1, int J, int V,
                        int ldl, float slope, void *stream);
size_t tsasr_joint_f32_bwd_workspace_bytes(int B, int U1, int J);
int tsasr_joint_f32_bwd(const float *dlogits, const float *enc, const float *dec, const float *W, float *denc, float *ddec, float *dW,
                        float *dbias, const int32_t *tlen, const int32_t *ulen, int B, int T, int U1, int J, int V, int ldl, float slope,
                        void *workspace, size_t workspace_bytes, void *stream);

size_t tsasr_joint_bwd_workspace_bytes(int B, int T, int U1, int J);
/* Backward of the above: denc[b,t,:], ddec[b,u,:] (io_dtype), dW[V,J], dbias[V] (fp32, OVERWRITTEN; while tsasr_reduce_defer(1)
 * is in force the two are parameter-gradient outputs like the others: written by tsasr_reduce_flush, workspace kept until then).
 * tlen/ulen (may be NULL = full) let the kernel skip the part of the lattice whose dlogits are zero. */
int tsasr_joint_bwd(const float *dlogits, const void *enc, const void *dec, const float *W,
                    void *denc, void *ddec, float *dW, float *dbias,
                    const int32_t *tlen, const int32_t *ulen,
                    int B, int T, int U1, int J, int V, int ldl, int io_dtype, float slope,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * RNN-T loss  (replaces SB/nnet/losses.py:29-87 `transducer_loss` -> torchaudio.functional.rnnt_loss,
 * call site train_librispeechmix_scratch.py:158-160; lattice math as SB/nnet/loss/transducer_loss.py:60-236)
 *   costs[b] = -log P(y_b | x_b)  (no /T), fused log-softmax over the V valid columns of each row.
 * fwd keeps alpha/beta/lse in `workspace` (same pointer must be handed to bwd);
 * bwd writes dlogits[b,t,u,v] = gscale[b] * d costs[b] / d logits  (zeros outside the lattice / v >= V).
 * ------------------------------------------------------------------------------------------ */
long long tsasr_rnnt_loss_error_word_offset(int B, int T, int U1);   /* split lattices (long targets, small batches): byte offset of the time-out word (1 = timed out), -1 = none */
size_t tsasr_rnnt_loss_workspace_bytes(int B, int T, int U1);
int tsasr_rnnt_loss_fwd(const float *logits, const int32_t *targets, int ldt, const int32_t *tlen,
                        const int32_t *ulen, float *costs, int B, int T, int U1, int V, int ldl, int blank,
                        void *workspace, size_t workspace_bytes, void *stream);
int tsasr_rnnt_loss_bwd(const float *logits, const int32_t *targets, int ldt, const int32_t *tlen,
                        const int32_t *ulen, const float *gscale /* [B] */, float *dlogits,
                        int B, int T, int U1, int V, int ldl, int blank,
                        const void *workspace, size_t workspace_bytes, void *stream);
/* Lab / test switch, no reference counterpart: how the next tsasr_rnnt_loss_* calls lay out and walk the alpha / beta lattice (-1 = the
 * default of each): skew 0 / 1 / 2 = frame-major planes always / anti-diagonal ("skewed") planes when a lattice runs on more than one
 * wave / always; waves = cap on the waves of a one-workgroup lattice; mc = 0 never split a lattice over workgroups, 1 / 2 / 4 = split
 * whenever the blocks fit on the chip, that many columns per thread. Changes tsasr_rnnt_loss_workspace_bytes; the forward and backward of
 * one loss must run under the same plan. Every plan returns the same bits (tests/test_rnnt_gpu.py). */
void tsasr_rnnt_lattice_plan(int skew, int waves, int mc);


/* ------------------------------------------------------------------------------------------
 * Row kernels of the Conformer block (HBM-bound, one pass each, fp32 math, deterministic column reductions).
 * LayerNorm: replaces torch.nn.LayerNorm / SB/nnet/normalization.py:172-223 as used at
 *   SB/lobes/models/transformer/Conformer.py:73,93,194-217 and SB/lobes/models/convolution.py (norm over [F,C]);
 *   act_slope >= 0 fuses the LeakyReLU that follows it (Conformer.py:93-95, convolution.py ConvBlock); < 0 = none.
 * x,y: [M, D] io_dtype; gamma,beta,dgamma,dbeta: fp32 [D]; mean,rstd: fp32 [M] (saved for backward). D % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
int tsasr_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                        long long M, int D, float eps, float act_slope, int io_dtype, void *stream);
size_t tsasr_layernorm_bwd_workspace_bytes(long long M, int D);
int tsasr_layernorm_bwd(const void *dy, const void *x, const float *gamma, const float *beta, const float *mean,
                        const float *rstd, void *dx, float *dgamma, float *dbeta, long long M, int D, float act_slope,
                        int io_dtype, void *workspace, size_t workspace_bytes, void *stream);
/* ... with dx = LayerNorm_bwd(dy) + dadd: the gradient that reached x along a residual path is summed in the same pass (rows of at
 * most 2048 bf16 / 1024 fp32 elements). */
int tsasr_layernorm_bwd_add(const void *dy, const void *dadd, const void *x, const float *gamma, const float *beta, const float *mean,
                            const float *rstd, void *dx, float *dgamma, float *dbeta, long long M, int D, float act_slope,
                            int io_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Dropout masks are a pure function of (seed + *seed_dev, element index); seed_dev (device uint64, may be NULL) is advanced once per
 * training step by tsasr_seed_advance so that a captured hipGraph still draws fresh masks every replay. */
int tsasr_seed_advance(unsigned long long *seed_dev, unsigned long long increment, void *stream);

/* y = dropout_p(act(x + bias))  - the Linear-bias + activation() + Dropout chain of PositionalwiseFeedForward
 * (SB/nnet/attention.py:820-836). bias may be NULL; act_slope < 0 = no activation; the dropout mask is a pure function
 * of (seed, element index) and is regenerated by the backward (never stored). dbias may be NULL. N % 8 == 0. */
int tsasr_bias_act_dropout_fwd(const void *x, const float *bias, void *y, long long M, int N, float act_slope, float p,
                               unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream);
size_t tsasr_colpart_workspace_bytes(long long M, int N);
int tsasr_bias_act_dropout_bwd(const void *dy, const void *y, void *dx, float *dbias, long long M, int N, float act_slope,
                               float p, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                               size_t workspace_bytes, void *stream);

/* out = res + alpha * timemask(dropout_p(x + bias)) - the "Dropout -> 0.5*x + residual" and
 * "Dropout -> masked_fill_(pad) -> + residual" tails of ConformerEncoderLayer / ConvolutionModule
 * (Conformer.py:113-114,239-259). rows = [B, Trows] flattened; valid_lens (int32 [B], may be NULL) zeroes frames
 * t >= valid_lens[b] of the x branch. res may be NULL. Backward gives dx (dres = dout needs no kernel). */
int tsasr_dropout_add_fwd(const void *x, const float *bias, const void *res, void *out, long long M, int N, float alpha,
                          float p, unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows,
                          int io_dtype, void *stream);
int tsasr_dropout_add_bwd(const void *dout, void *dx, float *dbias, long long M, int N, float alpha, float p,
                          unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows, int io_dtype,
                          void *workspace, size_t workspace_bytes, void *stream);
/* out = dropout_p2( res + alpha * timemask(dropout_p(x + bias)) ): the residual tail of a front-end ConvBlock together with the block's
 * outer Dropout (speechbrain/lobes/models/convolution.py:260-266) in one pass. Backward: dres = dropout_p2'(dout) (written; also the
 * gradient of `res`), dx = alpha * timemask * dropout_p'(dres); dres is required when p2 > 0. Both masks are counter-based
 * (seed / seed2 + *seed_dev) and regenerated in the backward. */
int tsasr_dropout_add2_fwd(const void *x, const float *bias, const void *res, void *out, long long M, int N, float alpha, float p,
                           unsigned long long seed, float p2, unsigned long long seed2, const unsigned long long *seed_dev,
                           const int32_t *valid_lens, int Trows, int io_dtype, void *stream);
int tsasr_dropout_add2_bwd(const void *dout, void *dx, void *dres, float *dbias, long long M, int N, float alpha, float p,
                           unsigned long long seed, float p2, unsigned long long seed2, const unsigned long long *seed_dev,
                           const int32_t *valid_lens, int Trows, int io_dtype, void *workspace, size_t workspace_bytes, void *stream);


/* ------------------------------------------------------------------------------------------
 * Conformer convolution-module core: z = LeakyReLU(LayerNorm(depthwise_conv1d_K(GLU(y2 + b2)))) on channels-last rows.
 * Replaces SB/lobes/models/transformer/Conformer.py:101-115 between the two pointwise GEMMs: bottleneck bias + nn.GLU
 * (:76-82), depthwise nn.Conv1d(D,D,K,groups=D) 'same' or causal pad+chomp (:68-71,84-93,108-110), after_conv LayerNorm +
 * activation (:95-97).  y2 [B,T,2D] io_dtype = bottleneck GEMM output without bias; b2 [2D] or NULL; conv_w [D,K] (the
 * [D,1,K] parameter); conv_b, gamma, beta [D]. D % 8 == 0, D <= 2048, K in {31,15,7,3}. c_save [B,T,D], mean/rstd [B*T] are
 * written by fwd and read by bwd. bwd: dparams fp32 = [dgamma D | dbeta D | dconv_b D | db2 2D | dconv_w D*K], overwritten.
 * ------------------------------------------------------------------------------------------ */
int tsasr_convmod_fwd(const void *y2, const float *b2, const float *conv_w, const float *conv_b, const float *gamma,
                      const float *beta, void *z, void *c_save, float *mean, float *rstd, int B, int T, int D, int K, int causal,
                      float eps, float slope, int io_dtype, void *stream);
size_t tsasr_convmod_bwd_workspace_bytes(int B, int T, int D, int K);
int tsasr_convmod_bwd(const void *dz, const void *y2, const float *b2, const float *conv_w, const float *gamma, const float *beta,
                      const void *c_save, const float *mean, const float *rstd, void *dy2, float *dparams, int B, int T, int D,
                      int K, int causal, float slope, int io_dtype, void *workspace, size_t workspace_bytes, void *stream);


/* Front-end block 2 as IMPLICIT GEMMs (csrc/gemm.hip conv_s2_*): the 3x3 stride-2 Conv2d + 1x1 stride-2 residual conv of a ConvBlock with
 * C_in > 1 (SB/lobes/models/convolution.py:178-266, SB/nnet/CNN.py:629-711) without the [P, 9*Ci] patch matrix: the ring kernels' loader
 * waves gather the patch rows from x [B,T,F,Ci] (bf16, channels-last) with the padding rule folded into the source address. Wm [128, 9*Ci] =
 * the 3x3 filter as [Co, (kt, kf, ci)], W2 [128, Ci] (bf16); b1 / b2 fp32 [128]; y1, y2 [B,T',F',128] bf16. Co == 128, Ci in {64, 128}.
 * Filter gradients: dWm [128, 9*Ci], dW2 [128, Ci] fp32 (overwritten; split-K slabs in the workspace, summed in fixed order) from dy1, dy2
 * [P,128] and x. Data gradient: dx [B,T,F,Ci] bf16 (every element written) as ONE gathered GEMM launch over classes of input pixels (which
 * taps and output positions reach a pixel depends on its parity and on the reflected border rows) - no [P, 9*Ci] gradient matrix, no inverse
 * gather pass (tsasr_frontend_col2im remains the fp32 / other-shape path). The plan (classes + launch order of their 128-pixel tiles)
 * depends on (B, T, F, causal) only: tsasr_conv3x3s2_dgrad_plan fills a HOST buffer of tsasr_conv3x3s2_dgrad_plan_bytes (0 = unsupported
 * shape), the caller uploads it once and passes the device copy. */
int tsasr_conv3x3s2_fwd(const void *x, const void *Wm, const float *b1, const void *W2, const float *b2, void *y1, void *y2, int B, int T, int F,
                        int Ci, int Co, int causal, void *stream);
size_t tsasr_conv3x3s2_wgrad_workspace_bytes(int B, int T, int F, int Ci);
int tsasr_conv3x3s2_wgrad(const void *dy1, const void *dy2, const void *x, float *dWm, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                          void *workspace, size_t workspace_bytes, void *stream);
/* the same with the 3x3 gradient in the reference's parameter layout dW1 [128][Ci][kF][kT] (torch Conv2d weight, SB/nnet/CNN.py:629-711): added to
 * the parameter's gradient as it is */
int tsasr_conv3x3s2_wgrad_filters(const void *dy1, const void *dy2, const void *x, float *dW1, float *dW2, int B, int T, int F, int Ci, int Co, int causal,
                                  void *workspace, size_t workspace_bytes, void *stream);
size_t tsasr_conv3x3s2_dgrad_plan_bytes(int B, int T, int F, int causal);
int tsasr_conv3x3s2_dgrad_plan(int B, int T, int F, int causal, void *plan_host, size_t plan_bytes);
int tsasr_conv3x3s2_dgrad(const void *dy1, const void *dy2, const void *Wm, const void *W2, void *dx, int B, int T, int F, int Ci, int Co, int causal,
                          const void *plan_dev, size_t plan_bytes, void *stream);
/* ------------------------------------------------------------------------------------------
 * Convolutional front-end (SB/lobes/models/convolution.py:103-266, Conv2d.forward SB/nnet/CNN.py:629-711): stride-2 3x3
 * conv ('same' = reflect padding, or causal = (2,0) zero pad on time, (1,1) on frequency) + 1x1 stride-2 residual conv.
 * Tensors are the reference's [B,T,F,C]. Block 1 (C_in = 1) is a direct kernel producing both branches; for wider inputs
 * the taps are gathered (im2col, column order (kt,kf,c)) so that the contraction is a plain library GEMM, and col2im is the
 * deterministic inverse gather-sum for the input gradient (dR = gradient of the 1x1 branch w.r.t. its sub-sampled input).
 * ------------------------------------------------------------------------------------------ */
int tsasr_frontend_out_len(int n);
int tsasr_frontend_c1_fwd(const void *x, const float *w1, const float *b1, const float *w2, const float *b2, void *y1, void *y2,
                          int B, int T, int F, int C, int causal, int io_dtype, void *stream);
size_t tsasr_frontend_c1_bwd_workspace_bytes(int C);
int tsasr_frontend_c1_bwd(const void *x, const void *dy1, const void *dy2, float *dparams, int B, int T, int F, int C, int causal,
                          int io_dtype, void *workspace, size_t workspace_bytes, void *stream);
int tsasr_frontend_im2col(const void *x, void *A, int B, int T, int F, int C, int causal, int io_dtype, void *stream);
int tsasr_frontend_col2im(const void *dA, const void *dR, void *dx, int B, int T, int F, int C, int causal, int io_dtype, void *stream);

/* One whole ConvBlock per direction (SB/lobes/models/convolution.py:187-266: convs = Conv2d 3x3 s2 -> LayerNorm([F',C]) -> LeakyReLU ->
 * Dropout; reduce_conv = Conv2d 1x1 s2 -> LayerNorm([F',C]); out = Dropout(convs(x) + reduce_conv(x))):
 *   out = Drop_p_outer( LN_r(r) + Drop_p_inner( LeakyReLU_slope( LN_y(y) ) ) ),   (y, r) = the two convolution outputs.
 * x != NULL (block 1, C_in = 1): (y, r) are computed on the fly from the features x [B,T,F] with filters w1 [C,1,3,3] (kernel axes (F,T)),
 *   b1, w2 [C], b2, and recomputed in the backward: nothing but `out` and 4 floats of statistics per (b, t') reach memory.
 * x == NULL (wider blocks): (y, r) = y1, y2 [B*T', F', C] as produced by the im2col GEMMs; the backward writes their gradients dy1, dy2.
 * T, F are always the block's INPUT sizes (T' = (T-1)/2+1, F' likewise). g1/be1, g2/be2 fp32 [F'*C] = LayerNorm affine of the 3x3 / 1x1
 * branch. stats fp32 [B*T'][4] = (mean_y, rstd_y, mean_r, rstd_r). Dropout masks are f(seed + *seed_dev, element index of out).
 * dparams fp32 (tsasr_frontend_block_dparams floats, overwritten): [dw1 C*9 | db1 C | dw2 C | db2 C] (x != NULL only), then
 * [dg1 | dbe1 | dg2 | dbe2] (F'*C each). Supported: tsasr_frontend_block_supported(F', C) (C = 128, F' <= 40). */
int tsasr_frontend_block_supported(int Fo, int C);
int tsasr_frontend_block_fwd(const void *x, const void *y1, const void *y2, const float *w1, const float *b1, const float *w2,
                             const float *b2, const float *g1, const float *be1, const float *g2, const float *be2, void *out,
                             float *stats, int B, int T, int F, int C, int causal, float slope, float eps, float p_inner,
                             unsigned long long seed_inner, float p_outer, unsigned long long seed_outer,
                             const unsigned long long *seed_dev, int io_dtype, void *stream);
size_t tsasr_frontend_block_dparams(int Fo, int C, int with_conv);
size_t tsasr_frontend_block_bwd_workspace_bytes(int Fo, int C, int with_conv);
int tsasr_frontend_block_bwd(const void *x, const void *y1, const void *y2, const void *dout, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *g1, const float *be1, const float *g2,
                             const float *stats, void *dy1, void *dy2, float *dparams, int B, int T, int F, int C, int causal,
                             float slope, float p_inner, unsigned long long seed_inner, float p_outer,
                             unsigned long long seed_outer, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                             size_t workspace_bytes, void *stream);


/* ------------------------------------------------------------------------------------------
 * Fused relative-position multi-head self-attention: replaces the body of RelPosMHAXL.forward between in_proj and out_proj,
 * SB/nnet/attention.py:586-633 (q+u/q+v, matrix_ac, matrix_bd + rel_shift :468-483, 1/sqrt(embed_dim) scale, -inf masks,
 * softmax, dropout, .V) without any T x T tensor in HBM.
 *   qkv [B,T,H,3*Dh] (per head Q|K|V, attention.py:549-553), pk [2T-1, H*Dh] = linear_pos(pos_embs),
 *   bias_u/bias_v fp32 [H*Dh] = the (Dh,H) parameters' storage read as [H,Dh] (a view, attention.py:586-592),
 *   key_lens int32 [B] or NULL (keys j >= key_lens[b] are masked), causal != 0 masks j > i.
 *   out [B,T,H*Dh]; lse fp32 [B,H,T] (log-sum-exp of the scaled scores; needed by the backward). Dh <= 64.
 * ------------------------------------------------------------------------------------------ */
size_t tsasr_relpos_attn_lds_bytes(void);
int tsasr_relpos_attn_fwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                          void *out, float *lse, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                          unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream);
/* Same with a caller-owned workspace: long sequences in small batches (B * H * T/128 < 512 workgroups, T > 1024) are split along the keys
 * across workgroups and merged by a second launch. workspace may be NULL (no split). */
size_t tsasr_relpos_attn_fwd_workspace_bytes(int B, int T, int H);
int tsasr_relpos_attn_fwd_ws(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                             void *out, float *lse, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                             unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace,
                             size_t workspace_bytes, void *stream);
size_t tsasr_relpos_attn_bwd_workspace_bytes(int B, int T, int H);
/* Backward: dqkv [B,T,H,3*Dh] fully written; d_bias_u/d_bias_v fp32 [H*Dh] ([H,Dh] reading of the parameter storage); dpk [2T-1, H*Dh]
 * (io_dtype, fully written) = gradient of pk = linear_pos(pos_embs). The workspace holds P_d and scale*dS as [B,H,T,ceil64(T)] tensors
 * between the query-major pass (which computes them) and the key-major pass (dK, dV = two contractions over the queries), the
 * q + pos_bias_v rows and the per-utterance-group partial sums of the d(pk) pass. */
int tsasr_relpos_attn_bwd(const void *qkv, const void *pk, const float *bias_u, const float *bias_v, const int32_t *key_lens,
                          const void *out, const void *dout, const float *lse, void *dqkv, void *dpk, float *d_bias_u,
                          float *d_bias_v, int B, int T, int H, int Dh, float scale, int causal, float pdrop,
                          unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *workspace, size_t workspace_bytes,
                          void *stream);
/* Deferred d(pk) (csrc/attention.hip): the d(pk) pass of tsasr_relpos_attn_bwd feeds only the weight gradient of linear_pos. While
 * tsasr_relpos_dpk_defer(1) is in force the call QUEUES the pass (workspace, key_lens and dpk of each call stay alive and untouched;
 * dpk is not written yet) and tsasr_relpos_dpk_flush runs all queued passes as one launch + one launch for the sums of their partials
 * (same blocks and order of sums as the per-call launches: bit-identical dpk). table_host: pinned host memory, table_dev: device memory,
 * both >= tsasr_relpos_dpk_table_bytes(tsasr_relpos_dpk_pending()); the tsasr_wgrad_flush protocol under stream capture. */
int tsasr_relpos_dpk_defer(int on);
int tsasr_relpos_dpk_pending(void);
size_t tsasr_relpos_dpk_table_bytes(int max_jobs);
int tsasr_relpos_dpk_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream);
void tsasr_relpos_dpk_discard(void);
/* Dropout keep-bits handed from the forward to the backward (an optimisation of the pair above, not a change of their results): with
 * bits = tsasr_relpos_attn_keepbits_bytes(B, T, H) > 0 bytes of device memory set by tsasr_relpos_attn_keepbits right before
 * tsasr_relpos_attn_fwd[_ws] (bf16, Dh = 64, 2 <= T <= 256, pdrop > 0) the forward stores the mask bits it hashed, and set again right
 * before the matching tsasr_relpos_attn_bwd the backward reads them (one 16-byte load per query row) instead of hashing every element
 * again. The setting is consumed by the next forward / backward call; without it both hash - the same bits. */
size_t tsasr_relpos_attn_keepbits_bytes(int B, int T, int H);
void tsasr_relpos_attn_keepbits(void *bits);

/* ------------------------------------------------------------------------------------------
 * Global-norm clipping + AdamW over the flat parameter arena: replaces SB/core.py:1082-1093
 * (torch.nn.utils.clip_grad_norm_ -> torch.optim.AdamW.step) with two launches; the norm stays on the device.
 * p, g, m, v: flat fp32 [n]; p_bf16 (may be NULL): bf16 shadow of p rewritten in the same pass; hyper: DEVICE float[3] = {lr, 1-beta1^t, 1-beta2^t} (graph-capturable Noam schedule).
 * norm_out (may be NULL): DEVICE float = L2 norm of g before clipping. skipped_out: NULL = the reference's behaviour on a non-finite norm (SB/core.py:1072-1093: the step is
 * applied - clip_grad_norm_'s factor is NaN / 0 - and only the non-finite loss is counted); a DEVICE float = such a step is SKIPPED (parameters and moments untouched) and
 * counted there (+= 1): the build's `skip_nonfinite_step: True` option.
 * ------------------------------------------------------------------------------------------ */
/* dst[i] += src[i] for `count` small fp32 vectors in one launch; table (DEVICE) = [count src ptrs][count dst ptrs][count int32 lengths]. */
int tsasr_accumulate_many(const void *table, int count, void *stream);
size_t tsasr_clip_adamw_workspace_bytes(void);
int tsasr_clip_adamw_step(float *p, void *p_bf16, const float *g, float *m, float *v, const float *hyper, float *norm_out, float *skipped_out,
                          long long n, float beta1, float beta2, float eps, float weight_decay, float max_norm, void *workspace,
                          size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with the operand layouts of a Linear layer (replaces the library GEMMs behind torch.nn.functional.linear on
 * SB/nnet/attention.py:549-553,581-583,635,820-836, Conformer.py:76-82,98, SB/nnet/linear.py:64-78):
 *   C[M,N] (+)= op(A)[M,K] . op(B)[K,N];  transA=0: A [M,K], 1: A [K,M];  transB=0: B [N,K] (a Linear weight), 1: B [K,N].
 *   out_dtype TSASR_BF16 | TSASR_F32; accumulate (fp32 only; 2 = the slab sum may be deferred, see tsasr_reduce_defer): C += result, used to add weight gradients straight into the
 *   fp32 gradient arena (long inner dimensions are split into fp32 slabs in `workspace`, summed in a fixed order).
 * ------------------------------------------------------------------------------------------ */
void tsasr_gemm_set_ring(int on);                 /* 1 (default): LDS-DMA ring main loop for long inner dimensions; 2: whenever K % 64 == 0; 0: register-staged loop (A/B tests) */
void tsasr_gemm_set_lab_floor(int floor_mode);   /* lab (tools/gemm_bench.py --floors): -1 off; 0/1/2 = round-3 four-wave loop / its DMA ring alone / its MFMA part alone */
void tsasr_gemm_set_plan(int tile, int splits);   /* A/B tests only: force macro-tile (0 = 128x128, 1 = 128x64, 2 = 64x64, -1 = automatic) and split-K */
size_t tsasr_gemm_bf16_workspace_bytes(int M, int N, int K, int out_dtype);
int tsasr_gemm_bf16(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                    int transA, int transB, int out_dtype, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* Same GEMM (bf16 out) with a fused elementwise epilogue - the Linear + activation() + Dropout of PositionalwiseFeedForward
 * (SB/nnet/attention.py:820-836) without a separate pass over the [M, d_ffn] hidden activation:
 *   epi_mode 1: C = dropout_p(LeakyReLU_slope(A.B + bias[n]));   epi_mode 2 (its backward on the dgrad GEMM):
 *   C = (A.B) * keep(m,n)/(1-p) * LeakyReLU'(y[m,n]) and dbias[n] = column sums of C.  Masks: counter-based, (seed + *seed_dev, m*N+n). */
/* fp32 GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): the Linear layers of the PARITY mode - the reference's default precision is
 * fp32 (conformer-t_scratch.yaml:88; SB/nnet/linear.py:64-78, attention.py:549-553, 581-583, 820-836, Conformer.py:76-82, 98), the
 * benchmarked step computes them in bf16. C[M,N] (+)= op(A) . op(B), operands and sums in fp32, layouts as tsasr_gemm_bf16; accumulate
 * != 0: C += result. Any shape and stride. */
int tsasr_gemm_f32(const float *A, const float *B, float *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int transA,
                   int transB, int accumulate, void *stream);
/* nbatch products with one shared left operand in ONE launch: C_i [M,N] = A [M,K] . B_i^T, B_i = btab[i] (a DEVICE array of device
 * pointers to bf16 [N,K] matrices, row stride ldb), C_i = C + i * c_batch elements. The 12 + 6 `linear_pos` projections of the one
 * positional table (SB/nnet/attention.py:433, 560: every RelPosMHAXL layer projects the same pos_embs). K % 64 == 0, N % 8 == 0. */
int tsasr_gemm_bf16_nt_batched(const void *A, const void *const *btab, void *C, int M, int N, int K, long long lda, long long ldb,
                               long long ldc, long long c_batch, int nbatch, void *stream);
size_t tsasr_gemm_bf16_fused_workspace_bytes(int M, int N);
/* `mask` (may be NULL; only where tsasr_gemm_bf16_fused_mask_ok(M, N, K) and transA = transB = 0): uint16 [M][N/8], one word per 8
 * consecutive outputs - bits 0-7 the dropout keep-bits, bits 8-15 "the stored activation is negative". The mode-1 call writes it; the
 * mode-2 call given the same words reads them INSTEAD of y and of re-hashing the keep-bits (y must still be passed): M*N/4 bytes read
 * instead of 2*M*N. Same results bit for bit. */
int tsasr_gemm_bf16_fused_mask_ok(int M, int N, int K);
int tsasr_gemm_bf16_fused(const void *A, const void *B, void *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                          int transA, int transB, int epi_mode, const float *bias, const void *y, long long ldy, float slope, float p,
                          unsigned long long seed, const unsigned long long *seed_dev, float *dbias, void *mask, void *workspace,
                          size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * LSTM cell steps of the prediction network (replaces torch.nn.LSTM behind SB/nnet/RNN.py:244-278; gate order i,f,g,o).
 * gates [B,U,H,4] fp32 - GATE-MINOR: the gates (i,f,g,o) of a unit are one float4, i.e. column 4*k + g of a [B*U, 4H] matrix, so
 * the caller permutes the rows of W_ih (and of W_hh if it adds h_{t-1} W_hh^T itself with tsasr_gemm_bf16 accumulate) from the
 * reference's gate-major order: pre-activations in, ACTIVATED gates out; c [B,U,H] fp32; h [B,U,H] io_dtype. One launch per time step; step t reads step t-1.
 * bwd: dout [B,U,H] (gradient of the layer output), dh_rec [B,H] fp32 (= dgates_{t+1} . W_hh, ignored at t = U-1),
 * dc_io [B,H] fp32 carried between steps, dgates [B,U,4H] io_dtype out (operand of the dh / dW GEMMs).
 * ------------------------------------------------------------------------------------------ */
int tsasr_lstm_cell_fwd(float *gates, float *c, void *h, int B, int U, int H, int t, int io_dtype, void *stream);
/* Fused variants: recurrent product (MFMA, K split over the waves) + cell update in one launch per step; whh bf16 [4H,H], whhT bf16 [H,4H]. */
int tsasr_lstm_step_fwd(float *gates, float *c, void *h, const void *whh, int B, int U, int H, int t, int io_dtype, void *stream);
int tsasr_lstm_step_bwd(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, float *dc_io, int B, int U,
                        int H, int t, int io_dtype, void *stream);
int tsasr_lstm_cell_bwd(const float *gates, const float *c, const void *dout, const float *dh_rec, float *dc_io, void *dgates, int B,
                        int U, int H, int t, int io_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * A1 Fbank: replaces SB/lobes/features.py:130-147 = STFT (SB/processing/features.py:134-178; n_fft 512, periodic Hamming 512,
 * center=True zero padding, one-sided) + spectral_magnitude power=1 (:317-348) + Filterbank (:482-552) + _amplitude_to_DB
 * (:683-704; floor at per-utterance max - top_db over ALL frames incl. padding).  wav [B,L] fp32 -> out [B,T=1+L/hop,n_mels].
 * A2 InputNormalization(norm_type="sentence"): SB/processing/features.py:1012-1025,1107-1132 (mean / unbiased std over the
 * first lens[b] frames per feature bin, applied to the whole padded row; eps clamps the std).
 * ------------------------------------------------------------------------------------------ */
size_t tsasr_fbank_workspace_bytes(int B, int T, int n_mels);
int tsasr_fbank_fwd(const float *wav, const float *window, const float *melmat, void *out, int B, int L, int T, int n_mels, int hop,
                    float top_db, float amin, int out_dtype, void *workspace, size_t workspace_bytes, void *stream);
int tsasr_sentence_norm_fwd(const void *x, const int32_t *lens, void *y, int B, int T, int F, float eps, int in_dtype, int out_dtype,
                            void *stream);

/* Fused seam between two Conformer sub-blocks: s = res + alpha*timemask(dropout_p(x + bias)); y = LayerNorm(s)
 * (Dropout -> [0.5*]x + residual -> masked_fill_ -> nn.LayerNorm; Conformer.py:113-114,194-217,239-259) and its backward
 * (dres = LN_bwd(dy) + dout; dx = alpha*timemask*dropmask/(1-p)*dres; dgamma, dbeta, dbias). D % 8 == 0, D <= 2048. */
int tsasr_add_layernorm_fwd(const void *x, const float *bias, const void *res, void *s, void *y, float *mean, float *rstd,
                            const float *gamma, const float *beta, long long M, int D, float alpha, float p, unsigned long long seed,
                            const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows, float eps, int io_dtype,
                            void *stream);
size_t tsasr_add_layernorm_bwd_workspace_bytes(long long M, int D);
int tsasr_add_layernorm_bwd(const void *dy, const void *dout, const void *s, const float *gamma, const float *mean, const float *rstd,
                            void *dres, void *dx, float *dgamma, float *dbeta, float *dbias, long long M, int D, float alpha, float p,
                            unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens, int Trows,
                            int io_dtype, void *workspace, size_t workspace_bytes, void *stream);
/* The same seam with the GEMM that produces x in front of it, in one launch: x = A[M,K] . W[N,K]^T (bf16, never written) for N = K = 256 -
 * the attention's out_proj (SB/nnet/attention.py:549-553) and the convolution module's last point-wise convolution (Conformer.py:76-98)
 * followed by `x + skip` / `x + conv(..)` and the LayerNorm that reads the sum (Conformer.py:243-259). (s, y, mean, rstd) are bit-identical
 * to tsasr_gemm_bf16 + tsasr_add_layernorm_fwd; the backward is tsasr_add_layernorm_bwd + the GEMM's. lda / ldw: row strides in elements. */
int tsasr_linear_add_layernorm_ok(long long M, int N, int K, long long lda, long long ldw);
int tsasr_linear_add_layernorm_fwd(const void *A, long long lda, const void *W, long long ldw, const float *bias, const void *res, void *s,
                                   void *y, float *mean, float *rstd, const float *gamma, const float *beta, long long M, int N, int K,
                                   float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev,
                                   const int32_t *valid_lens, int Trows, float eps, void *stream);
/* Two LayerNorms in a row with the residual tail in front - norm2 of a Conformer layer and the next layer's first LayerNorm (or the
 * encoder's final norm): Conformer.py:194-217,259, models/conformer.py:223-233. (s, y, z) bit-identical to tsasr_add_layernorm_fwd +
 * tsasr_layernorm_fwd; the backward takes dz and (optionally) the gradients reaching y and s along other paths. */
int tsasr_add_layernorm2_fwd(const void *x, const float *bias, const void *res, void *s, void *y, void *z, float *mean, float *rstd,
                             float *mean2, float *rstd2, const float *gamma, const float *beta, const float *gamma2, const float *beta2,
                             long long M, int D, float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev,
                             const int32_t *valid_lens, int Trows, float eps, float eps2, int io_dtype, void *stream);
size_t tsasr_add_layernorm2_bwd_workspace_bytes(long long M, int D);
int tsasr_add_layernorm2_bwd(const void *dz, const void *dy, const void *dout, const void *s, const float *gamma, const float *beta,
                             const float *gamma2, const float *mean, const float *rstd, const float *mean2, const float *rstd2, void *dres,
                             void *dx, float *dgamma, float *dbeta, float *dbias, float *dgamma2, float *dbeta2, long long M, int D,
                             float alpha, float p, unsigned long long seed, const unsigned long long *seed_dev, const int32_t *valid_lens,
                             int Trows, int io_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Whole-sequence LSTM recurrences (all U steps of tsasr_lstm_step_fwd / _bwd). bf16, H in {256, 512}, B <= 256: one persistent
 * launch per direction (workgroups exchange h_t / dG_t through write-through stores and an arrival counter); otherwise a loop of
 * the per-step kernels. Replaces the time loop inside torch.nn.LSTM (speechbrain/nnet/RNN.py:244-278). */
/* Input projection of the predictor when its Embedding is one-hot and frozen (speechbrain/nnet/embedding.py:76-95 consider_as_one_hot,
 * train_librispeechmix_scratch.py:117-119 `embedding` -> `decoder`): gates [B,U,H,4] fp32 = b_ih + b_hh + the token's column of w_ih
 * [4H, I] (fp32), i.e. F.embedding + x . W_ih^T + biases of torch.nn.LSTM without the gather, the casts and the GEMM. tokens int64 [B,U];
 * token k -> column k-1 above `blank`, k below it, blank -> zeros. xp (may be NULL): bf16 [B*U, Ip] one-hot rows + ones in columns I, I+1
 * (I + 2 <= Ip <= H): the operand the backward's weight-gradient GEMM contracts with. */
int tsasr_lstm_onehot_gates(const long long *tokens, const float *w_ih, const float *b_ih, const float *b_hh, float *gates, void *xp, int B,
                            int U, int H, int I, int Ip, int blank, void *stream);
int tsasr_lstm_seq_persistent(int B, int H, int io_dtype);   /* 1: one persistent launch per direction on this device for this shape */
size_t tsasr_lstm_seq_workspace_bytes(int B, int U, int H);   /* first 256 bytes: uint32 {arrival counter, error word} per batch group; a non-zero error word = an inter-workgroup wait timed out (outputs poisoned with NaN) */
int tsasr_lstm_seq_fwd(float *gates, float *c, void *h, const void *whh, int B, int U, int H, int io_dtype, void *workspace,
                       size_t workspace_bytes, void *stream);
int tsasr_lstm_seq_bwd(const float *gates, const float *c, const void *dout, void *dgates, const void *whhT, int B, int U, int H,
                       int io_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Transposed bf16 copies of a list of row-major matrices (GEMM weights -> k-contiguous operands of the input-gradient GEMMs), one
 * launch. jobs: DEVICE int32 [njobs][5] = {src offset, dst offset (elements), rows, cols, first 64x64 tile index}. */
int tsasr_transpose_many_bf16(const void *src_base, void *dst_base, const void *jobs, int njobs, int ntiles, void *stream);

/* Grouped weight gradients (csrc/wgrad.hip): replaces, for every Linear / kernel-size-1 Conv1d weight of the path, the reference's
 * autograd AccumulateGrad of dW = dy^T . x (SB/nnet/linear.py:64-78, SB/nnet/attention.py:549-553,820-836,
 * SB/lobes/models/transformer/Conformer.py:76-98) and the per-weight split-K launches of round 1. tsasr_wgrad_queue only records
 *   dW[M,N] (fp32, row stride ldc) += dy[K,M]^T . x[K,N]      (bf16 row-major, K = tokens; M, N, ld_dy, ld_x multiples of 8)
 * - dy, x and dW must stay alive and unmodified until tsasr_wgrad_flush runs every queued job in ONE launch of 256x256 output
 * tiles (no split along K: each tile owns its piece of dW, plain read-add-store, bit-reproducible). Two queued jobs must not
 * target overlapping dW. Job table protocol as tsasr_reduce_flush (pinned host + device table; under stream capture the caller
 * uploads the table after the capture). */
int tsasr_wgrad_queue(const void *dy, const void *x, float *dW, int M, int N, int K, long long ld_dy, long long ld_x, long long ldc);
int tsasr_wgrad_pending(void);
size_t tsasr_wgrad_table_bytes(int max_jobs);
int tsasr_wgrad_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream);
void tsasr_wgrad_discard(void);
/* The NEXT tsasr_wgrad_flush runs with `slots` LDS-DMA slots (2: 64 KB of LDS per workgroup, for a launch that should share the CUs
 * with another stream's kernels; 0: default 128 KB). One-shot. */
void tsasr_wgrad_next_flush_slots(int slots);
/* The NEXT tsasr_wgrad_flush launches at most `wgs` workgroups that walk the tiles persistently (0: one workgroup per tile). One-shot. */
void tsasr_wgrad_next_flush_wgs(int wgs);

/* Recipe glue on the device (csrc/misc.hip), each ONE launch instead of a chain of tiny library kernels:
 * tsasr_mean_pool_*: masked mean over time of the speaker encoder's output (train_librispeechmix_scratch.py:52-64);
 * tsasr_abs_lengths: relative -> absolute lengths (round half to even: models/conformer.py:272, SB/nnet/losses.py:58-59; floor:
 *   SB/nnet/RNN.py:35; ceil clamped: train_librispeechmix_scratch.py:54-58) for up to 8 vectors at once;
 * tsasr_count_nonfinite: the non-finite-loss counter of SB/core.py:1115-1150 kept on the device. */
int tsasr_mean_pool_fwd(const void *x, const float *rel, void *out, int B, int T, int D, int io_dtype, void *stream);
int tsasr_mean_pool_bwd(const void *dout, const float *rel, void *dx, int B, int T, int D, int io_dtype, void *stream);
int tsasr_abs_lengths(const float *const *rel, int *const *out, const int *dim, const int *mode, int n, int B, void *stream);
/* Speaker-embedding injection modes `sum` / `prod` (models/conformer.py:247-253; mode 0 / 1): out [B,T,D] = src [B,T,D] (+ | *) spk [B,1,D],
 * D % 8 == 0; backward in one pass over dout: dsrc [B,T,D] (mode 0: may be NULL - it equals dout), dspk [B,1,D] (sums over t in a fixed order). (`cat` = column ranges of
 * tsasr_gemm_bf16*, `cross_attention` = tsasr_attn_f32_* between GEMMs.) */
int tsasr_inject_fwd(const void *src, const void *spk, void *out, int B, int T, int D, int mode, int io_dtype, void *stream);
int tsasr_inject_bwd(const void *dout, const void *src, const void *spk, void *dsrc, void *dspk, int B, int T, int D, int mode, int io_dtype,
                     void *stream);
/* ------------------------------------------------------------------------------------------
 * Greedy transducer search on the device (SB/decoders/transducer.py:138-218, transducer_greedy_decode): one launch decodes the batch,
 * one persistent workgroup per utterance; predictor = embedding table -> one-layer LSTM -> Linear, joiner = LeakyReLU(enc + pn) -> Linear
 * head; at most one symbol per frame, the predictor advances only on a non-blank. enc [B,T,J] (io_dtype); emb fp32 [n_emb, E], E <= 64;
 * LSTM weights in torch layout (w_ih [4H,E], w_hh [4H,H], gate order i,f,g,o); w_proj [J,H]; w_head [V,J], V <= 63; matrices in `wdtype`
 * (TSASR_F32 | TSASR_BF16), biases fp32 or NULL; H, J multiples of 4. preds int32 [B,T]: symbol emitted at frame t or -1;
 * logp_sum fp32 [B]: sum of the emitted symbols' log-probabilities.
 * ------------------------------------------------------------------------------------------ */
int tsasr_greedy_decode(const void *enc, const float *emb, const void *w_ih, const void *w_hh, const float *b_ih, const float *b_hh,
                        const void *w_proj, const float *b_proj, const void *w_head, const float *b_head, int *preds, float *logp_sum,
                        int B, int T, int J, int H, int E, int V, int blank, float slope, int io_dtype, int wdtype, void *stream);

int tsasr_count_nonfinite(const float *x, int n, int *counter, void *stream);

/* Direct RCCL gradient all-reduce over xGMI (csrc/comm.hip): replaces the NCCL calls behind the reference's per-module
 * DistributedDataParallel reducers (SB/core.py:1464-1484; `no_sync` :1585-1615) and SB/utils/distributed.py:123-201's process-group
 * collectives for the gradient path. One communicator per process (one process per GPU). librccl is dlopen()ed
 * (tsasr_allreduce_load; NULL = "librccl.so" on the loader path - pass the copy PyTorch has loaded); rank 0 makes the 128-byte unique
 * id (tsasr_allreduce_unique_id), the caller ships it to the other ranks, every rank calls tsasr_allreduce_init. A bucket is reduced
 * IN PLACE on `stream` (sum or average over the ranks), asynchronously and graph-capturably; "wait" is a stream join by the caller. */
int tsasr_allreduce_load(const char *librccl_path);
int tsasr_allreduce_unique_id(void *host_id128);
int tsasr_allreduce_init(const void *host_id128, int nranks, int rank);
int tsasr_allreduce_ready(void);   /* number of ranks of the communicator, 0 = none */
int tsasr_allreduce_bucket(void *buf, size_t count, int dtype, int average, void *stream);
int tsasr_allreduce_destroy(void);

/* Batched deterministic reductions of partial gradient rows / split-K slabs (csrc/reduce.hip). While tsasr_reduce_defer(1) is in
 * force the parameter-gradient outputs of the *_bwd entry points (dgamma, dbeta, dbias, conv-module dparams, fused-GEMM dbias)
 * and of tsasr_gemm_bf16(accumulate = 2) are only QUEUED: their workspaces and outputs must stay alive and untouched until
 * tsasr_reduce_flush runs all queued jobs in one launch (table_host: pinned host memory, table_dev: device memory, both at least
 * tsasr_reduce_table_bytes(tsasr_reduce_pending()) bytes; must outlive a captured graph). While `stream` is being captured the flush
 * fills table_host only: the caller copies it to table_dev once, after the capture has ended, so a replayed graph carries no memcpy
 * node. Replaces ~310 small launches per step. */
int tsasr_reduce_defer(int on);
int tsasr_reduce_pending(void);
/* Drops every queued reduction without running it and switches deferral off (error path of a step that raised half-way). */
void tsasr_reduce_discard(void);
size_t tsasr_reduce_table_bytes(int max_jobs);
int tsasr_reduce_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream);
/* Only the jobs whose partial rows were produced on `stream` (complete in its order): the main stream reduces its share while a
 * forked stream still runs its part of backward; the rest goes with the final tsasr_reduce_flush. Needs its own table pair. */
int tsasr_reduce_flush_stream(void *table_host, void *table_dev, size_t table_bytes, void *stream);

/* SpecAugment on the normalised features (speechbrain/lobes/augment.py:32-201, applied at train_librispeechmix_scratch.py:91-94;
 * recipe settings conformer-t_scratch.yaml:132-142). tsasr_specaug_draw produces every random number of one call on the device
 * (params: tsasr_specaug_params_words(B, n_freq_mask, n_time_mask) int32 = {c, w, freq widths [B][nf], freq starts, time widths
 * [B][nt], time starts}; c ~ U[window, T-window), w ~ U[c-window, c+window)+1, width ~ U[lo, hi), start ~ U[0, max(1, D - max width
 * over the batch)); window == 0 or T - window <= window: c = w = 0 = no warp). tsasr_specaug_apply: y = masks(warp(x)) for x, y
 * [B,T,F] (y != x): bicubic (cubic convolution, align_corners) resize of [0,c) -> [0,w) and [c,T) -> [w,T) along time, then frequency
 * masks filled with mean(warped) and time masks filled with mean(frequency-masked) (0 when replace_with_zero). At most 8 masks per
 * axis. No host synchronisation anywhere (the reference syncs once per mask axis on mask_len.max()). */
size_t tsasr_specaug_params_words(int B, int n_freq_mask, int n_time_mask);
int tsasr_specaug_draw(int32_t *params, int B, int T, int F, int window, int n_freq_mask, int f_lo, int f_hi, int n_time_mask, int t_lo,
                       int t_hi, unsigned long long seed, const unsigned long long *seed_dev, void *stream);
size_t tsasr_specaug_workspace_bytes(void);
int tsasr_specaug_apply(const void *x, void *y, const int32_t *params, int B, int T, int F, int n_freq_mask, int n_time_mask,
                        int replace_with_zero, int io_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Speed perturbation = polyphase windowed-sinc resampling of the waveform (speechbrain/processing/speech_augmentation.py:435-820,
 * applied at train_librispeechmix_scratch.py:82-85; recipe: 16 kHz -> 15.2 / 16 / 16.8 kHz). x [B,L] fp32 -> y [B,n_out] with
 * n_out = tsasr_resample_out_len(L, orig, new); weights [P,W] fp32 / first [P] int32 (device) are the filter bank and first input
 * index per phase (P = new/gcd phases, stride = orig/gcd input samples per unit), built by the host as Resample._indices_and_weights
 * does. Output n = q*P + i is sum_j weights[i][j] * x[first[i] + q*stride + j] with zeros outside the signal. */
long long tsasr_resample_out_len(long long n_in, int orig_freq, int new_freq);
int tsasr_resample_fwd(const float *x, float *y, const float *weights, const int32_t *first, int B, int L, int n_out, int P, int stride,
                       int W, void *stream);

/* out[c] (+)= sum_m x[m][c], x [M, N] io_dtype (N % 8 == 0, N <= 2048): bias gradient of a GEMM-shaped layer (front-end convolutions,
 * speechbrain/nnet/CNN.py:629-676). The final sum over per-workgroup partials goes through the deferrable batched reduction. */
size_t tsasr_colsum_workspace_bytes(long long M, int N);
int tsasr_colsum(const void *x, float *out, long long M, int N, int accumulate, int io_dtype, void *workspace, size_t workspace_bytes,
                 void *stream);

/* ------------------------------------------------------------------------------------------
 * Single-layer LSTM in EXACT fp32 arithmetic with an optional initial state (csrc/lstm_f32.hip): the `compute_dtype: fp32` parity mode of
 * the predictor (torch.nn.LSTM behind SB/nnet/RNN.py:170-278) and the step-wise predictor calls of the searchers
 * (SB/decoders/transducer.py:246-353,411-466: one token, carried (h, c)). Gate order i, f, g, o; x [B,U,I], w_ih [4H,I], w_hh [4H,H],
 * biases [4H] or NULL, h0 / c0 [B,H] or NULL (zeros). hs [B,U,H]; hn / cn [B,H] or NULL; cs [B,U,H] and gates [B,U,4H] (activated) are
 * what the backward needs (NULL when not training). Backward: dgates [B,U,4H] w.r.t. the gate pre-activations (the weight / bias / input
 * gradients are GEMMs over all (b,t): tsasr_gemm_f32), dh0 / dc0 optional. H <= 1024.
 * ------------------------------------------------------------------------------------------ */
int tsasr_lstm_f32_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *h0,
                       const float *c0, float *hs, float *cs, float *gates, float *hn, float *cn, int B, int U, int I, int H, void *stream);
int tsasr_lstm_f32_bwd(const float *dout, const float *dhn, const float *dcn, const float *gates, const float *cs, const float *c0,
                       const float *w_hh, float *dgates, float *dh0, float *dc0, int B, int U, int H, void *stream);

/* ------------------------------------------------------------------------------------------
 * Attention in EXACT fp32 arithmetic (csrc/attention_f32.hip; no matrix cores, no bf16 rounding of operands): the `compute_dtype: fp32`
 * parity mode of RelPosMHAXL (same reference lines as tsasr_relpos_attn_*: SB/nnet/attention.py:586-633, rel_shift :468-483) and, with
 * pk == NULL and Tq != Tk, the core of torch.nn.MultiheadAttention for the `cross_attention` speaker injection (models/conformer.py:263-266).
 * q / k / v are strided views: strides[9] (HOST, in elements) = {q batch, q row, q head, k batch, k row, k head, v batch, v row, v head}
 * (RelPosMHAXL's interleaved qkv [B,T,H,3*Dh]: {T*3D, 3D, 3Dh} with k = qkv + Dh, v = qkv + 2*Dh). pk [2*Tk-1, H*Dh] or NULL,
 * bias_u / bias_v fp32 [H*Dh] or NULL; key_lens int32 [B] or NULL; causal as tsasr_relpos_attn_fwd (C > 1: block-causal extension).
 * out [B,Tq,H*Dh] (io_dtype), lse fp32 [B,H,Tq]. Dropout bit of element ((b*H+h)*Tq+i)*Tk+j from (seed + *seed_dev). Dh <= 64.
 * Backward: dq / dk / dv through dstrides[9] (same meaning), dpk [2*Tk-1, H*Dh] when pk != NULL, d_bias_u / d_bias_v fp32 [H*Dh] or NULL
 * (queued while tsasr_reduce_defer(1) is in force, like every parameter-gradient output). Three deterministic passes, no atomics;
 * the workspace holds two fp32 [B,H,Tq,Tk] planes.
 * ------------------------------------------------------------------------------------------ */
int tsasr_attn_f32_fwd(const void *q, const void *k, const void *v, const long long *strides, const void *pk, const float *bias_u,
                       const float *bias_v, const int32_t *key_lens, void *out, float *lse, int B, int Tq, int Tk, int H, int Dh, float scale,
                       int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype, void *stream);
size_t tsasr_attn_f32_bwd_workspace_bytes(int B, int Tq, int Tk, int H, int Dh);
int tsasr_attn_f32_bwd(const void *q, const void *k, const void *v, const long long *strides, const void *pk, const float *bias_u,
                       const float *bias_v, const int32_t *key_lens, const void *out, const void *dout, const float *lse, void *dq, void *dk,
                       void *dv, const long long *dstrides, void *dpk, float *d_bias_u, float *d_bias_v, int B, int Tq, int Tk, int H, int Dh,
                       float scale, int causal, float pdrop, unsigned long long seed, const unsigned long long *seed_dev, int io_dtype,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * LibriSpeechMix mixture on the device (csrc/dataio.hip): the arithmetic of the reference's `audio_pipeline`
 * (train_librispeechmix_scratch.py:356-386), which the reference runs per utterance in DataLoader workers on the host.
 * src = the utterance's sources back to back (fp32, device); host_off [nsrc+1] (HOST) their element offsets; host_delay [nsrc] (HOST)
 * ceil(delay_j * sample_rate) in samples; rescale != 0 (gain_nontarget != 0): every non-target source is multiplied by
 * sqrt(ratio * mean(target^2) / mean(src_j^2)), ratio = (float)10^(gain_nontarget/10), each operation in fp32 as the reference's 0-dim
 * tensor expression (:361-369; the mean powers are fp64 sums rounded once - the reference's fp32 cascade sum depends on the host's SIMD
 * width); sources are shifted right by their delay, zero-padded to the longest and summed LEFT TO RIGHT in fp32 (:372-377);
 * out [tsasr_mix_sources_out_len(...)] = mixture[start : start + duration] (both in samples; clipped to the mixture as a Python slice).
 * workspace: tsasr_mix_sources_workspace_bytes() (the mean powers), only read when rescale != 0. At most 8 sources.
 * ------------------------------------------------------------------------------------------ */
size_t tsasr_mix_sources_workspace_bytes(void);
long long tsasr_mix_sources_out_len(const long long *host_off, const int *host_delay, int nsrc, long long start, long long duration);
int tsasr_mix_sources(const float *src, const long long *host_off, const int *host_delay, int nsrc, int target, float ratio, int rescale,
                      long long start, long long duration, float *out, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
