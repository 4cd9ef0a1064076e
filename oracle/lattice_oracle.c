/*
 * oracle/lattice_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-thread CPU restatement of the reference's lattice arithmetic.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product path (pytorch-asr_amd/) never does.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.  Arithmetic is fp32 with the same operation order as
 * the reference's torch ops (logsumexp = max + log(sum(exp(x - max)))); the one
 * fp64 function (oracle_path_logsumexp_f64) is the arbiter for rounding questions.
 *
 * Pinning: tests/test_oracle_golden.py checks these functions against outputs
 * of the imported reference (tests/golden/*.npz, made by
 * tests/golden/make_golden.py in the build container).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* torch.logsumexp over k values (fst_utils.py:297-319): max, then
 * log(sum(exp(x - max))) + max.  All-equal -inf never happens here because the
 * reference uses the finite sentinel -1e20 (fst_utils.py:347). */
static float lse(const float *x, int k)
{
    float m = x[0];
    for (int i = 1; i < k; ++i)
        if (x[i] > m) m = x[i];
    float s = 0.f;
    for (int i = 0; i < k; ++i) s += expf(x[i] - m);
    return m + logf(s);
}

/*
 * PathLogSumExp.forward (fst_utils.py:403-480).
 *   lp      [T,B,C]  time-major log-probs
 *   lens    [B]      active frames per utterance (sorted descending upstream)
 *   in-arc matrices  src_in/il_in/w_in [Bg,N,Kin], term [Bg,N]
 *   out-arc matrices dst_out/il_out/w_out [Bg,N,Kout]
 *   Bg in {1,B} (fst_utils.py:406-410 expands batch-1 graphs)
 * outputs: logZ [B], grad [T,B,C] (zeros outside active frames, :448),
 *          alphas [T,B,N] pre-update alphas (:437), logZ_bwd [B] (:476).
 * The reference's "shrinking active batch" loops (:434-443, :451-471) visit,
 * for utterance b, exactly frames 0..lens[b]-1 forward and lens[b]-1..0
 * backward, which is what the per-utterance loops below do.
 */
void oracle_path_logsumexp(const float *lp, int T, int B, int C,
                           const int32_t *lens,
                           const int32_t *src_in, const int32_t *il_in,
                           const float *w_in, const float *term,
                           const int32_t *dst_out, const int32_t *il_out,
                           const float *w_out,
                           int N, int Kin, int Kout, int Bg, float neg_inf,
                           float *logZ, float *grad, float *alphas,
                           float *logZ_bwd)
{
    int kmax = Kin > Kout ? Kin : Kout;
    if (N > kmax) kmax = N;
    float *alpha = (float *)malloc(sizeof(float) * N);
    float *next = (float *)malloc(sizeof(float) * N);
    float *tok = (float *)malloc(sizeof(float) * (size_t)N * kmax);
    memset(grad, 0, sizeof(float) * (size_t)T * B * C);          /* :448 */
    for (size_t i = 0; i < (size_t)T * B * N; ++i) alphas[i] = neg_inf; /* :428 */

    for (int b = 0; b < B; ++b) {
        int g = (Bg == 1) ? 0 : b;
        const int32_t *s_i = src_in + (size_t)g * N * Kin;
        const int32_t *l_i = il_in + (size_t)g * N * Kin;
        const float *ww_i = w_in + (size_t)g * N * Kin;
        const float *tm = term + (size_t)g * N;
        const int32_t *d_o = dst_out + (size_t)g * N * Kout;
        const int32_t *l_o = il_out + (size_t)g * N * Kout;
        const float *ww_o = w_out + (size_t)g * N * Kout;
        int len = lens[b];

        for (int n = 0; n < N; ++n) alpha[n] = neg_inf;           /* :424 */
        alpha[0] = 0.f;                                           /* :425 */
        for (int t = 0; t < len; ++t) {
            const float *row = lp + ((size_t)t * B + b) * C;
            memcpy(alphas + ((size_t)t * B + b) * N, alpha, sizeof(float) * N); /* :437 */
            for (int n = 0; n < N; ++n) {
                float *tk = tok;
                for (int k = 0; k < Kin; ++k) {                   /* :438-441 */
                    float v = ww_i[n * Kin + k];
                    v += alpha[s_i[n * Kin + k]];
                    v += row[l_i[n * Kin + k]];
                    tk[k] = v;
                }
                next[n] = lse(tk, Kin);                           /* :442 */
            }
            memcpy(alpha, next, sizeof(float) * N);
        }
        for (int n = 0; n < N; ++n) tok[n] = alpha[n] + tm[n];
        float lz = lse(tok, N);                                   /* :445 */
        logZ[b] = lz;

        float *beta = alpha; /* reuse */
        for (int n = 0; n < N; ++n) beta[n] = tm[n];              /* :447 */
        for (int t = len - 1; t >= 0; --t) {
            const float *row = lp + ((size_t)t * B + b) * C;
            float *grow = grad + ((size_t)t * B + b) * C;
            const float *al = alphas + ((size_t)t * B + b) * N;
            for (int n = 0; n < N; ++n) {
                float *tk = tok + (size_t)n * Kout;
                for (int k = 0; k < Kout; ++k) {                  /* :457-460 */
                    float v = ww_o[n * Kout + k];
                    v += beta[d_o[n * Kout + k]];
                    v += row[l_o[n * Kout + k]];
                    tk[k] = v;
                }
                next[n] = lse(tk, Kout);                          /* :461 */
            }
            for (int n = 0; n < N; ++n) {
                float a = al[n] - lz;                             /* :463-465 */
                for (int k = 0; k < Kout; ++k) {
                    float o = expf(tok[(size_t)n * Kout + k] + a); /* :466 */
                    grow[l_o[n * Kout + k]] += o;                 /* :468-470 */
                }
            }
            memcpy(beta, next, sizeof(float) * N);
        }
        if (logZ_bwd) {                                           /* :476 */
            for (int n = 0; n < N; ++n) tok[n] = (n == 0 ? 0.f : neg_inf) + beta[n];
            logZ_bwd[b] = lse(tok, N);
        }
    }
    free(alpha);
    free(next);
    free(tok);
}

/*
 * fp64 arbiter of oracle_path_logsumexp: the SAME recurrences and operation order
 * (fst_utils.py:424-471) with every intermediate in double (inputs are the fp32
 * tensors, widened).  Not a restatement of anything the reference runs — the
 * reference computes in fp32 — but the yardstick that says, where the fp32 oracle
 * and the GPU kernel differ, whose rounding it is: tests bound the kernel's error
 * against this by the fp32 oracle's own error against this.
 * outputs: logZ [B], grad [T,B,C], logZ_bwd [B], all double.
 */
static double lse64(const double *x, int k)
{
    double m = x[0];
    for (int i = 1; i < k; ++i)
        if (x[i] > m) m = x[i];
    double s = 0.0;
    for (int i = 0; i < k; ++i) s += exp(x[i] - m);
    return m + log(s);
}

void oracle_path_logsumexp_f64(const float *lp, int T, int B, int C,
                               const int32_t *lens,
                               const int32_t *src_in, const int32_t *il_in,
                               const float *w_in, const float *term,
                               const int32_t *dst_out, const int32_t *il_out,
                               const float *w_out,
                               int N, int Kin, int Kout, int Bg, float neg_inf,
                               double *logZ, double *grad, double *logZ_bwd)
{
    int kmax = Kin > Kout ? Kin : Kout;
    if (N > kmax) kmax = N;
    double *alpha = (double *)malloc(sizeof(double) * N);
    double *next = (double *)malloc(sizeof(double) * N);
    double *tok = (double *)malloc(sizeof(double) * (size_t)N * kmax);
    memset(grad, 0, sizeof(double) * (size_t)T * B * C);
    for (int b = 0; b < B; ++b) {
        int g = (Bg == 1) ? 0 : b;
        const int32_t *s_i = src_in + (size_t)g * N * Kin;
        const int32_t *l_i = il_in + (size_t)g * N * Kin;
        const float *ww_i = w_in + (size_t)g * N * Kin;
        const float *tm = term + (size_t)g * N;
        const int32_t *d_o = dst_out + (size_t)g * N * Kout;
        const int32_t *l_o = il_out + (size_t)g * N * Kout;
        const float *ww_o = w_out + (size_t)g * N * Kout;
        int len = lens[b];
        double *alphas = (double *)malloc(sizeof(double) * (size_t)(len > 0 ? len : 1) * N);

        for (int n = 0; n < N; ++n) alpha[n] = neg_inf;
        alpha[0] = 0.0;
        for (int t = 0; t < len; ++t) {
            const float *row = lp + ((size_t)t * B + b) * C;
            memcpy(alphas + (size_t)t * N, alpha, sizeof(double) * N);
            for (int n = 0; n < N; ++n) {
                for (int k = 0; k < Kin; ++k)
                    tok[k] = (double)ww_i[n * Kin + k] + alpha[s_i[n * Kin + k]] +
                             (double)row[l_i[n * Kin + k]];
                next[n] = lse64(tok, Kin);
            }
            memcpy(alpha, next, sizeof(double) * N);
        }
        for (int n = 0; n < N; ++n) tok[n] = alpha[n] + (double)tm[n];
        double lz = lse64(tok, N);
        logZ[b] = lz;

        double *beta = alpha;
        for (int n = 0; n < N; ++n) beta[n] = tm[n];
        for (int t = len - 1; t >= 0; --t) {
            const float *row = lp + ((size_t)t * B + b) * C;
            double *grow = grad + ((size_t)t * B + b) * C;
            const double *al = alphas + (size_t)t * N;
            for (int n = 0; n < N; ++n) {
                double *tk = tok + (size_t)n * Kout;
                for (int k = 0; k < Kout; ++k)
                    tk[k] = (double)ww_o[n * Kout + k] + beta[d_o[n * Kout + k]] +
                            (double)row[l_o[n * Kout + k]];
                next[n] = lse64(tk, Kout);
            }
            for (int n = 0; n < N; ++n) {
                double a = al[n] - lz;
                for (int k = 0; k < Kout; ++k)
                    grow[l_o[n * Kout + k]] += exp(tok[(size_t)n * Kout + k] + a);
            }
            memcpy(beta, next, sizeof(double) * N);
        }
        if (logZ_bwd) {
            for (int n = 0; n < N; ++n) tok[n] = (n == 0 ? 0.0 : (double)neg_inf) + beta[n];
            logZ_bwd[b] = lse64(tok, N);
        }
        free(alphas);
    }
    free(alpha);
    free(next);
    free(tok);
}

/*
 * path_reduction's alpha-only scan (fst_utils.py:349-397) with reduction
 * 'logsumexp' (viterbi=0) or torch.max (viterbi=1, :366-370), plus — for the
 * viterbi case — the per-frame ilabel on the best path, which is what
 * FSTDecoder.decode reads out of the autograd gradient
 * (advanced_decoder.py:546-554: grad is -1 at [t,b,ilabel of the arg-max arc]).
 * Ties: first maximum (torch.max on CPU returns the first maximal index).
 * best_il [T,B] int32, 0 for t >= lens[b] (grad row all-zero -> min index 0).
 */
void oracle_path_forward(const float *lp, int T, int B, int C,
                         const int32_t *lens,
                         const int32_t *src_in, const int32_t *il_in,
                         const float *w_in, const float *term,
                         int N, int K, int Bg, float neg_inf, int viterbi,
                         float *score, int32_t *best_il)
{
    float *alpha = (float *)malloc(sizeof(float) * N);
    float *next = (float *)malloc(sizeof(float) * N);
    float *tok = (float *)malloc(sizeof(float) * (N > K ? N : K));
    int32_t *bp = NULL;
    if (viterbi && best_il) {
        bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * N);
        memset(best_il, 0, sizeof(int32_t) * (size_t)T * B);
    }
    for (int b = 0; b < B; ++b) {
        int g = (Bg == 1) ? 0 : b;
        const int32_t *s_i = src_in + (size_t)g * N * K;
        const int32_t *l_i = il_in + (size_t)g * N * K;
        const float *ww = w_in + (size_t)g * N * K;
        const float *tm = term + (size_t)g * N;
        int len = lens[b];
        for (int n = 0; n < N; ++n) alpha[n] = neg_inf;           /* :377 */
        alpha[0] = 0.f;                                           /* :378 */
        for (int t = 0; t < len; ++t) {
            const float *row = lp + ((size_t)t * B + b) * C;
            for (int n = 0; n < N; ++n) {
                int arg = 0;
                for (int k = 0; k < K; ++k) {                     /* :387-390 */
                    float v = alpha[s_i[n * K + k]] + ww[n * K + k];
                    v += row[l_i[n * K + k]];
                    tok[k] = v;
                    if (v > tok[arg]) arg = k;
                }
                if (viterbi) {
                    next[n] = tok[arg];
                    if (bp) bp[(size_t)t * N + n] = arg;
                } else {
                    next[n] = lse(tok, K);                        /* :391 */
                }
            }
            memcpy(alpha, next, sizeof(float) * N);
        }
        int argn = 0;
        for (int n = 0; n < N; ++n) {                             /* :396 */
            tok[n] = alpha[n] + tm[n];
            if (tok[n] > tok[argn]) argn = n;
        }
        score[b] = viterbi ? tok[argn] : lse(tok, N);
        if (bp) {
            int st = argn;
            for (int t = len - 1; t >= 0; --t) {
                int k = bp[(size_t)t * N + st];
                best_il[(size_t)t * B + b] = l_i[st * K + k];
                st = s_i[st * K + k];
            }
        }
    }
    free(alpha);
    free(next);
    free(tok);
    free(bp);
}

/*
 * get_normalized_acts (ctc_losses.py:29-43).  acts [R, C] rows = T*B.
 * mode 0: log_softmax over all C classes (normalize_by_dim falsy, :41-42).
 * mode 1: C = S*S viewed [R,S,S], log_softmax over the LAST axis
 *         (normalize_by_dim=1 -> torch axis 3, :34-40): blocks of S contiguous.
 * mode 2: normalize_by_dim=... axis 2 of [R,S,S] (strided blocks) — kept for
 *         completeness of the view arithmetic (context_order=2, dim=0 is falsy
 *         in the reference so only dim=1 reaches this branch there).
 */
void oracle_log_softmax(const float *acts, long R, int C, int S, int mode,
                        float *out)
{
    for (long r = 0; r < R; ++r) {
        const float *x = acts + r * C;
        float *y = out + r * C;
        if (mode == 0) {
            float m = x[0];
            for (int c = 1; c < C; ++c) if (x[c] > m) m = x[c];
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(x[c] - m);
            float l = logf(s);
            for (int c = 0; c < C; ++c) y[c] = x[c] - m - l;
        } else if (mode == 1) {
            for (int blk = 0; blk < C / S; ++blk) {
                const float *xb = x + blk * S;
                float m = xb[0];
                for (int c = 1; c < S; ++c) if (xb[c] > m) m = xb[c];
                float s = 0.f;
                for (int c = 0; c < S; ++c) s += expf(xb[c] - m);
                float l = logf(s);
                for (int c = 0; c < S; ++c) y[blk * S + c] = xb[c] - m - l;
            }
        } else {
            for (int j = 0; j < S; ++j) {
                float m = x[j];
                for (int i = 1; i < C / S; ++i) if (x[i * S + j] > m) m = x[i * S + j];
                float s = 0.f;
                for (int i = 0; i < C / S; ++i) s += expf(x[i * S + j] - m);
                float l = logf(s);
                for (int i = 0; i < C / S; ++i) y[i * S + j] = x[i * S + j] - m - l;
            }
        }
    }
}
