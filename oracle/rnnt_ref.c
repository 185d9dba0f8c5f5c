/* CPU ORACLE for the RNN-T (transducer) loss  -- TEST INFRASTRUCTURE, not product code.
 *
 * The reference reaches this arithmetic through a third-party dependency that is NOT vendored:
 *   speechbrain/nnet/losses.py:72-79  ->  torchaudio.functional.rnnt_loss(logits, targets,
 *   input_lens, target_lens, blank=0, reduction="mean")   (torchaudio>=0.10, requirements.txt:13)
 * whose published algorithm (Graves 2012, "Sequence Transduction with RNNs", eq. 16-20; the
 * warp-transducer formulation torchaudio documents) is restated here in double precision:
 *   lp = log_softmax(logits)                                  (fused log-softmax)
 *   alpha[0,0]=0; alpha[t,u] = logaddexp(alpha[t-1,u]+lp[t-1,u,blank], alpha[t,u-1]+lp[t,u-1,y_u])
 *   beta[T-1,U]=lp[T-1,U,blank]; beta[t,u] = logaddexp(beta[t+1,u]+lp[t,u,blank], beta[t,u+1]+lp[t,u,y_{u+1}])
 *   cost = -(alpha[T-1,U] + lp[T-1,U,blank]) = -beta[0,0]     (NO division by T)
 *   dcost/dlogits[t,u,v] = softmax[v]*exp(alpha+beta-logP) - occupancy(t,u,v)
 * The same recurrences are spelled out by the reference's own (non-default) Numba kernels,
 * speechbrain/nnet/loss/transducer_loss.py:60-106 (alpha), :137-180 (beta), :211-236 (occupancies),
 * which differ only in dividing the cost by T (:104-106) and in differentiating w.r.t. log-probs.
 * Pinned by the reference's known-answer test tests/unittests/test_losses.py:109-152
 * (2.2478 = cost/T with T=2  <=>  cost = 4.4957 here); see tests/test_rnnt_oracle.py.
 *
 * Layout: logits[b][t][u][0..ldl) with the first V entries of each row meaningful (ldl >= V lets
 * the checker read the product's padded rows); targets[b][0..maxU); grads same layout as logits.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double logaddexp_d(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}

/* returns 0 on success; costs[b] = -log P(y_b | x_b); grads may be NULL */
int rnnt_ref_loss(const float *logits, int B, int maxT, int maxU1, int V, int ldl,
                  const int *targets, int ldt, const int *tlen, const int *ulen, int blank,
                  double *costs, float *grads, double *alpha_out, double *beta_out) {
    if (grads) memset(grads, 0, sizeof(float) * (size_t)B * maxT * maxU1 * ldl);
    double *lpb = (double *)malloc(sizeof(double) * (size_t)maxT * maxU1);  /* blank log-prob */
    double *lpe = (double *)malloc(sizeof(double) * (size_t)maxT * maxU1);  /* emit log-prob  */
    double *lse = (double *)malloc(sizeof(double) * (size_t)maxT * maxU1);
    double *al = (double *)malloc(sizeof(double) * (size_t)maxT * maxU1);
    double *be = (double *)malloc(sizeof(double) * (size_t)maxT * maxU1);
    if (!lpb || !lpe || !lse || !al || !be) return -1;
    for (int b = 0; b < B; ++b) {
        const int T = tlen[b], U = ulen[b]; /* U labels -> U+1 lattice columns */
        if (T < 1 || T > maxT || U < 0 || U + 1 > maxU1) return -2;
        const float *lg = logits + (size_t)b * maxT * maxU1 * ldl;
        const int *y = targets + (size_t)b * ldt;
        for (int t = 0; t < T; ++t)
            for (int u = 0; u <= U; ++u) {
                const float *r = lg + ((size_t)t * maxU1 + u) * ldl;
                double m = r[0];
                for (int v = 1; v < V; ++v) if (r[v] > m) m = r[v];
                double s = 0.0;
                for (int v = 0; v < V; ++v) s += exp((double)r[v] - m);
                double l = m + log(s);
                lse[t * maxU1 + u] = l;
                lpb[t * maxU1 + u] = (double)r[blank] - l;
                lpe[t * maxU1 + u] = (u < U) ? (double)r[y[u]] - l : -INFINITY;
            }
#define A(t, u) al[(t) * maxU1 + (u)]
#define Bt(t, u) be[(t) * maxU1 + (u)]
#define LB(t, u) lpb[(t) * maxU1 + (u)]
#define LE(t, u) lpe[(t) * maxU1 + (u)]
        for (int t = 0; t < T; ++t)
            for (int u = 0; u <= U; ++u) {
                if (t == 0 && u == 0) { A(0, 0) = 0.0; continue; }
                double no_emit = (t > 0) ? A(t - 1, u) + LB(t - 1, u) : -INFINITY;
                double emit = (u > 0) ? A(t, u - 1) + LE(t, u - 1) : -INFINITY;
                A(t, u) = logaddexp_d(no_emit, emit);
            }
        for (int t = T - 1; t >= 0; --t)
            for (int u = U; u >= 0; --u) {
                if (t == T - 1 && u == U) { Bt(t, u) = LB(t, u); continue; }
                double no_emit = (t < T - 1) ? Bt(t + 1, u) + LB(t, u) : -INFINITY;
                double emit = (u < U) ? Bt(t, u + 1) + LE(t, u) : -INFINITY;
                Bt(t, u) = logaddexp_d(no_emit, emit);
            }
        const double logp = A(T - 1, U) + LB(T - 1, U);
        costs[b] = -logp;
        if (alpha_out) memcpy(alpha_out + (size_t)b * maxT * maxU1, al, sizeof(double) * maxT * maxU1);
        if (beta_out) memcpy(beta_out + (size_t)b * maxT * maxU1, be, sizeof(double) * maxT * maxU1);
        if (grads) {
            float *g = grads + (size_t)b * maxT * maxU1 * ldl;
            for (int t = 0; t < T; ++t)
                for (int u = 0; u <= U; ++u) {
                    const float *r = lg + ((size_t)t * maxU1 + u) * ldl;
                    float *gr = g + ((size_t)t * maxU1 + u) * ldl;
                    const double ab = A(t, u) + Bt(t, u) - logp;      /* log occupancy of the node */
                    const double l = lse[t * maxU1 + u];
                    for (int v = 0; v < V; ++v) {
                        double val = exp(ab + (double)r[v] - l);      /* softmax[v] * occ(node) */
                        if (v == blank) {
                            if (t == T - 1 && u == U) val -= exp(A(t, u) + LB(t, u) - logp);
                            else if (t < T - 1) val -= exp(A(t, u) + LB(t, u) + Bt(t + 1, u) - logp);
                        }
                        if (u < U && v == y[u]) val -= exp(A(t, u) + LE(t, u) + Bt(t, u + 1) - logp);
                        gr[v] = (float)val;
                    }
                }
        }
    }
    free(lpb); free(lpe); free(lse); free(al); free(be);
    return 0;
}
