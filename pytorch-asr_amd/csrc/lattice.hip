// Lattice scans for gfx950: log-semiring forward-backward (PathLogSumExp,
// reference att_speech/fst_utils.py:400-488) and the alpha-only scan with
// logsumexp / max reduction + best-path read-out (fst_utils.py:322-397,
// modules/decoders/advanced_decoder.py:546-554).
//
// Mapping: one workgroup per utterance; alpha/beta live in LDS (double
// buffered, one s_barrier per frame); the per-frame gradient row is
// accumulated in LDS (ds_add_f32) and streamed out coalesced, so every
// [t,b,:] row of the dense gradient is written exactly once.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "../../include/asr_amd.h"

namespace {

using namespace asr;

struct FwbwParams {
    const float *lp;
    int T, B, C;
    const int32_t *lens;
    const int32_t *src_in, *il_in;
    const float *w_in, *term;
    const int32_t *dst_out, *il_out;
    const float *w_out;
    int N, Kin, Kout, Bg;
    float neg_inf;
    float *logZ, *grad, *logZ_bwd;
    float *alphas;  // workspace, [T+2,B,roundup(N,64)] for the meet-in-the-middle kernels
    unsigned *redo; // band kernel: running count of utterances redone by the fallback body, or null
    float gsign;    // the posteriors are written times this (+1; -1: the gradient of -logZ)
    const int32_t *ctc_labels, *ctc_label_lens;   // band kernel: the transcripts [B, ctc_lmax] the graphs
    int ctc_lmax;                                 // were built from (context order 1), or null
};

// KR > 0: every thread owns ONE state (N <= blockDim) and keeps its <= KR
// in-arcs / out-arcs in registers for the whole scan.  KR == 0: states are
// strided over the block and arcs are streamed from global memory (L2) every
// frame (large shared graphs such as the CTC-G denominator).
// b: utterance; alphas_b / astride: this utterance's workspace rows (row t at alphas_b +
// t * astride, N floats each) — [T,B,N] for the stand-alone kernel, the calling kernel's own
// region when used as its fallback
template <int KR>
__device__ __forceinline__ void lattice_fwbw_generic_body(const FwbwParams &p, float *smem, int b,
                                                          float *alphas_b, size_t astride) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int N = p.N, C = p.C, Kin = p.Kin, Kout = p.Kout;
    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    float *abuf = smem;                 // [2][Npad]
    float *row = smem + 2 * Npad;       // [2][Cpad]
    float *red = row + 2 * Cpad;        // [32]

    const int g = (p.Bg == 1) ? 0 : b;
    const int32_t *src_in = p.src_in + (size_t)g * N * Kin;
    const int32_t *il_in = p.il_in + (size_t)g * N * Kin;
    const float *w_in = p.w_in + (size_t)g * N * Kin;
    const float *term = p.term + (size_t)g * N;
    const int32_t *dst_out = p.dst_out + (size_t)g * N * Kout;
    const int32_t *il_out = p.il_out + (size_t)g * N * Kout;
    const float *w_out = p.w_out + (size_t)g * N * Kout;
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;          // lp / grad frame stride
    const float *lp_b = p.lp + (size_t)b * C;
    float *grad_b = p.grad + (size_t)b * C;
    const float half_inf = p.neg_inf * 0.5f;

    // rows past the utterance end are zeros (fst_utils.py:448)
    for (int t = len; t < p.T; ++t)
        for (int c = tid; c < C; c += NT) grad_b[(size_t)t * tstride + c] = 0.f;

    for (int n = tid; n < Npad; n += NT) abuf[n] = (n == 0) ? 0.f : p.neg_inf;
    for (int c = tid; c < 2 * Cpad; c += NT) row[c] = 0.f;

    // ---- register-resident arcs (KR > 0) ----
    constexpr int KA = KR > 0 ? KR : 1;
    int r_src[KA], r_il[KA];
    float r_w[KA];
    const bool own = tid < N;
    if constexpr (KR > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            bool v = own && k < Kin;
            r_src[k] = v ? src_in[tid * Kin + k] : 0;
            r_il[k] = v ? il_in[tid * Kin + k] : 0;
            r_w[k] = v ? w_in[tid * Kin + k] : p.neg_inf;
        }
    }
    __syncthreads();

    // ---------------- forward ----------------
    int cur = 0;
    for (int t = 0; t < len; ++t) {
        const float *a = abuf + cur * Npad;
        float *an = abuf + (cur ^ 1) * Npad;
        const float *lrow = lp_b + (size_t)t * tstride;
        float *arow = alphas_b + (size_t)t * astride;
        if constexpr (KR > 0) {
            if (own) {
                arow[tid] = a[tid];                      // alphas[t] = pre-update
                float v[KA];
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    v[k] = r_w[k] + a[r_src[k]] + lrow[r_il[k]];
                    m = fmaxf(m, v[k]);
                }
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KR; ++k) s += __expf(v[k] - m);
                an[tid] = m + __logf(s);
            }
        } else {
            for (int n = tid; n < N; n += NT) {
                arow[n] = a[n];
                Lse acc;
                acc.init();
                for (int k = 0; k < Kin; ++k) {
                    int i = n * Kin + k;
                    acc.add(w_in[i] + a[src_in[i]] + lrow[il_in[i]]);
                }
                an[n] = acc.value();
            }
        }
        cur ^= 1;
        __syncthreads();
    }

    // logZ = logsumexp_n(alpha + terminal)   (fst_utils.py:445)
    float logZ;
    {
        const float *a = abuf + cur * Npad;
        float m = -INFINITY;
        for (int n = tid; n < N; n += NT) m = fmaxf(m, a[n] + term[n]);
        m = block_max(m, red);
        float s = 0.f;
        for (int n = tid; n < N; n += NT) s += __expf(a[n] + term[n] - m);
        s = block_sum(s, red);
        logZ = m + __logf(s);
        if (tid == 0) p.logZ[b] = logZ;
    }
    __syncthreads();

    // ---------------- backward ----------------
    if constexpr (KR > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            bool v = own && k < Kout;
            r_src[k] = v ? dst_out[tid * Kout + k] : 0;
            r_il[k] = v ? il_out[tid * Kout + k] : 0;
            r_w[k] = v ? w_out[tid * Kout + k] : p.neg_inf;
        }
    }
    cur = 0;
    for (int n = tid; n < N; n += NT) abuf[n] = term[n];        // beta = terminal (:447)
    __syncthreads();

    int rcur = 0;
    for (int t = len - 1; t >= 0; --t) {
        const float *bt = abuf + cur * Npad;
        float *bn = abuf + (cur ^ 1) * Npad;
        float *rw = row + rcur * Cpad;
        float *rprev = row + (rcur ^ 1) * Cpad;
        const float *lrow = lp_b + (size_t)t * tstride;
        const float *arow = alphas_b + (size_t)t * astride;
        // flush the row finished in the previous step (frame t+1), re-zero it
        if (t + 1 < len) {
            float *gout = grad_b + (size_t)(t + 1) * tstride;
            for (int c = tid; c < C; c += NT) {
                gout[c] = rprev[c];
                rprev[c] = 0.f;
            }
        }
        if constexpr (KR > 0) {
            if (own) {
                float v[KA];
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    v[k] = r_w[k] + bt[r_src[k]] + lrow[r_il[k]];
                    m = fmaxf(m, v[k]);
                }
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KR; ++k) s += __expf(v[k] - m);
                bn[tid] = m + __logf(s);
                const float a = arow[tid] - logZ;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    if (r_w[k] > half_inf) {
                        float o = __expf(v[k] + a) * p.gsign;
                        if (o != 0.f) atomicAdd(&rw[r_il[k]], o);
                    }
                }
            }
        } else {
            for (int n = tid; n < N; n += NT) {
                Lse acc;
                acc.init();
                const float a = arow[n] - logZ;
                for (int k = 0; k < Kout; ++k) {
                    int i = n * Kout + k;
                    float w = w_out[i];
                    int il = il_out[i];
                    float v = w + bt[dst_out[i]] + lrow[il];
                    acc.add(v);
                    if (w > half_inf) {
                        float o = __expf(v + a) * p.gsign;
                        if (o != 0.f) atomicAdd(&rw[il], o);
                    }
                }
                bn[n] = acc.value();
            }
        }
        cur ^= 1;
        rcur ^= 1;
        __syncthreads();
    }
    if (len > 0) {
        float *rprev = row + (rcur ^ 1) * Cpad;
        for (int c = tid; c < C; c += NT) grad_b[c] = rprev[c];
    }
    if (p.logZ_bwd) {                                           // (:476)
        const float *bt = abuf + cur * Npad;
        float m = -INFINITY;
        for (int n = tid; n < N; n += NT)
            m = fmaxf(m, bt[n] + (n == 0 ? 0.f : p.neg_inf));
        m = block_max(m, red);
        float s = 0.f;
        for (int n = tid; n < N; n += NT)
            s += __expf(bt[n] + (n == 0 ? 0.f : p.neg_inf) - m);
        s = block_sum(s, red);
        if (tid == 0) p.logZ_bwd[b] = m + __logf(s);
    }
}

template <int KR>
__global__ void lattice_fwbw_kernel(FwbwParams p) {
    extern __shared__ float smem[];
    lattice_fwbw_generic_body<KR>(p, smem, blockIdx.x, p.alphas + (size_t)blockIdx.x * p.N,
                                  (size_t)p.B * p.N);
}

// ---------------------------------------------------------------------------
// Meet-in-the-middle forward-backward for small-degree graphs (the CTC
// numerator lattices: N <= 512 states, <= KR arcs per state).
//
// The alpha recurrence and the beta recurrence are independent, so the
// workgroup runs them CONCURRENTLY: threads [0,H) ("A group") scan alpha
// forward from frame 0, threads [H,2H) ("B group") scan beta backward from
// frame len-1; each owns one state and keeps its arcs in registers.  They meet
// at m = len/2, where logZ = logsumexp_n(alpha_m + beta_m) is known, and keep
// going: past the meeting point the A group has beta_{t+1} (stored by the B
// group on its way down) and the B group has alpha_t, so each turns its own
// arc tokens into occupancies on the fly.  The sequential depth is len steps
// instead of 2*len, with one s_barrier per step.
//
// Emissions lp[t,b,il] do not depend on the recurrence: they are fetched D
// frames ahead into a register ring (the barrier is a bare s_barrier +
// lgkmcnt(0), so the loads stay in flight across it).
//
// Workspace ws[t,b,n]: alpha_t[n] for t < m, beta_{t+1}[n] for t >= m.
// ---------------------------------------------------------------------------
#ifdef ASR_ABLATE_NOEMIT
#define EMIT_LOAD(t, k) (-1.f - 0.001f * (float)r_il[k])
#else
#define EMIT_LOAD(t, k) lp_b[(size_t)(t) * tstride + r_il[k]]
#endif
#ifdef ASR_ABLATE_NOWS
#define WS_STORE(t, val) do { if ((val) == 123.f) ws_b[0] = 0.f; } while (0)
#define WS_LOAD(t) (-3.f)
#else
#define WS_STORE(t, val) ws_b[(size_t)(t) * astride + n] = (val)
#define WS_LOAD(t) ws_b[(size_t)(t) * astride + n]
#endif
template <int KR, int D>
__device__ __forceinline__ void lattice_fwbw_mitm_body(const FwbwParams &p, float *smem) {
    const int b = blockIdx.x;
    const int H = blockDim.x >> 1;
    const int grp = threadIdx.x >= H ? 1 : 0;        // wave-uniform (H % 64 == 0)
    const int n = threadIdx.x - grp * H;
    const int N = p.N, C = p.C;
    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    float *sbuf = smem + grp * 2 * Npad;             // [2][Npad] alpha | beta
    float *row = smem + 4 * Npad + grp * 2 * Cpad;   // [2][Cpad] per group
    float *red = smem + 4 * Npad + 4 * Cpad;         // [32]

    const int g = (p.Bg == 1) ? 0 : b;
    const int K = grp ? p.Kout : p.Kin;
    const int32_t *oth = (grp ? p.dst_out : p.src_in) + (size_t)g * N * K;
    const int32_t *ilp = (grp ? p.il_out : p.il_in) + (size_t)g * N * K;
    const float *wp = (grp ? p.w_out : p.w_in) + (size_t)g * N * K;
    const float *term = p.term + (size_t)g * N;
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const int m = len >> 1;          // meeting frame
    const int P = len - m;           // iterations per phase (>= m)
    const size_t tstride = (size_t)p.B * C;
    const float *lp_b = p.lp + (size_t)b * C;
    float *grad_b = p.grad + (size_t)b * C;
    // workspace rows are [B, H] like the state-labelled kernel's (a mixed batch has
    // both kinds of workgroup in one launch: their regions must not overlap)
    const size_t astride = (size_t)p.B * H;
    float *ws_b = p.alphas + (size_t)b * H;
    const float half_inf = p.neg_inf * 0.5f;
    const bool own = n < N;

    for (int t = len; t < p.T; ++t)                  // fst_utils.py:448
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            grad_b[(size_t)t * tstride + c] = 0.f;

    int r_oth[KR], r_il[KR];
    float r_w[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        bool v = own && k < K;
        r_oth[k] = v ? oth[n * K + k] : 0;
        r_il[k] = v ? ilp[n * K + k] : 0;
        r_w[k] = v ? wp[n * K + k] : p.neg_inf;
    }
    // alpha_0 = [0, neg_inf, ...]; beta_len = terminal
    for (int i = n; i < Npad; i += H)
        sbuf[i] = grp ? (i < N ? term[i] : p.neg_inf) : (i == 0 ? 0.f : p.neg_inf);
    for (int c = n; c < 2 * Cpad; c += H) row[c] = 0.f;
    __syncthreads();

    // frame visited by this group at iteration j of phase ph (-1: idle)
    //   A: phase 0 -> j (j < m)        phase 1 -> m + j
    //   B: phase 0 -> len-1-j          phase 1 -> m-1-j (j < m)
    auto frame_of = [&](int ph, int j) -> int {
        if (grp == 0) return ph == 0 ? (j < m ? j : -1) : m + j;
        return ph == 0 ? len - 1 - j : (j < m ? m - 1 - j : -1);
    };

    float e[D][KR];
    float wsr[D];
    int cur = 0;

    // ================= phase 0: plain scans, store alpha_t / beta_{t+1} =====
#pragma unroll
    for (int u = 0; u < D; ++u) {
        int t = u < P ? frame_of(0, u) : -1;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            e[u][k] = t >= 0 ? EMIT_LOAD(t, k) : 0.f;
    }
    for (int j0 = 0; j0 < P; j0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int j = j0 + u;
            if (j < P) {
                const int t = frame_of(0, j);
                if (t >= 0 && own) {
                    const float *s = sbuf + cur * Npad;
                    WS_STORE(t, s[n]);
                    float v[KR];
                    float mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KR; ++k) {
                        v[k] = r_w[k] + s[r_oth[k]] + e[u][k];
                        mx = fmaxf(mx, v[k]);
                    }
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < KR; ++k) sum += __expf(v[k] - mx);
                    sbuf[(cur ^ 1) * Npad + n] = mx + __logf(sum);
                }
                const int tn = (j + D < P) ? frame_of(0, j + D) : -1;
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    e[u][k] = tn >= 0 ? EMIT_LOAD(tn, k) : 0.f;
                if (t >= 0) cur ^= 1;
                __syncthreads();
            }
        }
    }

    // ================= meeting point ========================================
    // prime phase-1 rings first so the loads fly during the reduction
#pragma unroll
    for (int u = 0; u < D; ++u) {
        int t = u < P ? frame_of(1, u) : -1;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            e[u][k] = t >= 0 ? EMIT_LOAD(t, k) : 0.f;
        wsr[u] = (t >= 0 && own) ? WS_LOAD(t) : 0.f;
    }
    float logZm;
    // both groups need to know which buffer the other group ended on: A made
    // m updates, B made P updates, both starting at buffer 0.
    const int curA = m & 1, curB = P & 1;
    {
        const float *al = smem + curA * Npad;
        const float *be = smem + 2 * Npad + curB * Npad;
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < N; i += blockDim.x) mx = fmaxf(mx, al[i] + be[i]);
        mx = block_max(mx, red);
        float sum = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) sum += __expf(al[i] + be[i] - mx);
        sum = block_sum(sum, red);
        logZm = mx + __logf(sum);
    }
    __syncthreads();

    // ================= phase 1: scans + occupancies =========================
    int rc = 0;
    for (int j0 = 0; j0 < P; j0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int j = j0 + u;
            if (j < P) {
                const int t = frame_of(1, j);
                if (t >= 0) {
                    // flush the row finished one step ago, re-zero it
                    if (j > 0) {
                        const int tp = grp == 0 ? t - 1 : t + 1;
                        float *gout = grad_b + (size_t)tp * tstride;
                        float *rp = row + (rc ^ 1) * Cpad;
                        for (int c = n; c < C; c += H) {
                            gout[c] = rp[c];
                            rp[c] = 0.f;
                        }
                    }
                    if (own) {
                        const float *s = sbuf + cur * Npad;
                        float *rw = row + rc * Cpad;
                        float v[KR];
                        float mx = -INFINITY;
#pragma unroll
                        for (int k = 0; k < KR; ++k) {
                            v[k] = r_w[k] + s[r_oth[k]] + e[u][k];
                            mx = fmaxf(mx, v[k]);
                        }
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < KR; ++k) sum += __expf(v[k] - mx);
                        sbuf[(cur ^ 1) * Npad + n] = mx + __logf(sum);
                        const float a = wsr[u] - logZm;
#pragma unroll
                        for (int k = 0; k < KR; ++k) {
                            if (r_w[k] > half_inf) {
                                float o = __expf(v[k] + a) * p.gsign;
#ifndef ASR_ABLATE_NOATOMIC
                                if (o != 0.f) atomicAdd(&rw[r_il[k]], o);
#else
                                if (o == 123.f) rw[0] = o;
#endif
                            }
                        }
                    }
                }
                const int tn = (j + D < P) ? frame_of(1, j + D) : -1;
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    e[u][k] = tn >= 0 ? EMIT_LOAD(tn, k) : 0.f;
                wsr[u] = (tn >= 0 && own) ? WS_LOAD(tn) : 0.f;
                if (t >= 0) { cur ^= 1; rc ^= 1; }
                __syncthreads();
            }
        }
    }
    // flush each group's last row: A ends on frame len-1, B on frame 0
    {
        const int steps = grp == 0 ? P : m;
        if (steps > 0) {
            const int tl = grp == 0 ? len - 1 : 0;
            float *gout = grad_b + (size_t)tl * tstride;
            const float *rp = row + (rc ^ 1) * Cpad;
            for (int c = n; c < C; c += H) gout[c] = rp[c];
        }
    }
    // logZ = logsumexp_n(alpha_len + terminal) (fst_utils.py:445); A made len
    // updates in total.  logZ_bwd from beta_0 (fst_utils.py:476).
    {
        const float *al = smem + (len & 1) * Npad;
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < N; i += blockDim.x) mx = fmaxf(mx, al[i] + term[i]);
        mx = block_max(mx, red);
        float sum = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) sum += __expf(al[i] + term[i] - mx);
        sum = block_sum(sum, red);
        if (threadIdx.x == 0) p.logZ[b] = mx + __logf(sum);
    }
    if (p.logZ_bwd) {
        const float *be = smem + 2 * Npad + (len & 1) * Npad;
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            mx = fmaxf(mx, be[i] + (i == 0 ? 0.f : p.neg_inf));
        mx = block_max(mx, red);
        float sum = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            sum += __expf(be[i] + (i == 0 ? 0.f : p.neg_inf) - mx);
        sum = block_sum(sum, red);
        if (threadIdx.x == 0) p.logZ_bwd[b] = mx + __logf(sum);
    }
}

template <int KR, int D>
__global__ __launch_bounds__(1024) void lattice_fwbw_mitm_kernel(FwbwParams p) {
    extern __shared__ float smem[];
    lattice_fwbw_mitm_body<KR, D>(p, smem);
}

// ---------------------------------------------------------------------------
// Fast path for STATE-LABELLED graphs: every valid in-arc of a state carries
// the same input label.  True for all CTC lattices of the reference (a state
// of compose(decoding_fst, chain) is "the last emitted class", fst_utils.py:
// 679-835) and checked per utterance at kernel entry; a workgroup whose graph
// fails the check runs the generic per-arc-label body instead (same launch).
//
// With one label per state
//   alpha_{t+1}[n] = lp_t[label n] + LSE_k(w_k + alpha_t[src_k])
//   beta_t[n]      = LSE_k(w_k + (beta_{t+1} + lp_t[label .])[dst_k])
//   d logZ / d lp_t[c] = sum_{n: label n = c} exp(alpha_{t+1}[n] + beta_{t+1}[n] - logZ)
// i.e. ONE emission fetch and ONE posterior exp per state and frame instead of
// one per arc.  Same meet-in-the-middle schedule as lattice_fwbw_mitm_kernel.
// All scores are kept in log2 units so the recurrences use the raw v_exp_f32 /
// v_log_f32 (no range-reduction code); results are converted back on output.
//
// Per-class posterior sums (FL == 1, C <= H): the recurrence reads its sources
// through LDS pointers, so the assignment of states to lanes is free.  At kernel
// entry the states are counting-sorted BY LABEL (position = start of the label's
// run + rank within it); states of one class then sit in consecutive lanes and
// d logZ / d lp_t[c] is a SEGMENTED sum over lanes: four masked row_shr steps
// (1, 2, 4, 8) and two masked row broadcasts (lane 15 -> next row, lane 31 ->
// rows 2-3) leave a class's total over the wave in the last lane of its run,
// which stores it to the frame's gradient row in LDS — six VALU operations and
// one plain ds_write per lane and frame, where a scatter with LDS float adds
// costs ~4 cycles per ACTIVE lane (~100 states with a non-blank label per frame).
// Only classes whose run crosses a wave boundary (the blank of a CTC chain, at
// most one more per boundary) add their per-wave pieces with ds_add_f32.
// FL == 0 (C > H: bandwidth-bound, the row flush dominates) keeps one
// unconditional ds_add per lane with the wave's shared label reduced by DPP.
//
// ws[t,b,j] (log2 units, j = lane position): alpha_{t+1} for t < m, beta_{t+1} for t >= m.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    int x = __builtin_bit_cast(int, v);
    int y = __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, y);
}

// sum over the 64 lanes, result valid in every lane
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror  -> 16-lane row sums
    int x = __builtin_bit_cast(int, v);
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

// one step of a segmented inclusive scan: v += mask * v[lane picked by the DPP control]
// (lanes without a source — outside the row, or disabled by ROWS — add 0)
template <int CTRL, int ROWS>
__device__ __forceinline__ float seg_add(float v, float mask) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, true);
    return fmaf(__builtin_bit_cast(float, y), mask, v);
}

// the six steps of the segmented scan as v_fmac_f32 with a DPP source operand (hipcc emits
// v_mov_b32_dpp + v_fmac_f32 and a zeroing v_mov for the broadcasts: 17 instructions; here
// 6 + the wait states a DPP read of a just-written VGPR needs).  Lanes without a source
// (row start, rows a broadcast does not reach) are not written, i.e. add nothing.
__device__ __forceinline__ float seg_scan6(float v, float m1, float m2, float m4, float m8,
                                           float mb15, float mb31) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v)
        : "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mb15), "v"(mb31));
    return v;
}

#ifdef ASR_SL_NOBAR
#define SL_BARRIER() do {} while (0)
#else
#define SL_BARRIER() __syncthreads()
#endif
#define ASR_L2E 1.4426950408889634f
#define ASR_LN2 0.6931471805599453f

// waves_per_eu(6): <= 80 VGPRs, so THREE 8-wave workgroups fit a CU (two with the 89 the
// compiler takes otherwise): batches of 513..768 utterances stay one wave of workgroups
template <int K, int D, int FL>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
void lattice_fwbw_sl_kernel(FwbwParams p) {
    // FL == 1: C <= H: states sorted by label, per-class sums by segmented scan, the
    //          per-step row flush is one store per lane;
    // FL == 0: runtime flush loop (large C, bandwidth-bound regime), LDS float adds.
    extern __shared__ float smem[];
    typedef unsigned int u32;
    const int b = blockIdx.x;
    const int H = blockDim.x >> 1;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >= (unsigned)H ? 1 : 0);
    const int n = threadIdx.x - grp * H;
    const int N = p.N, C = p.C;
    const int Cpad = (C + 3) & ~3;
    float *sbuf = smem + grp * 2 * H;                 // [2][H] alpha | beta~
    float *row = smem + 4 * H + grp * 3 * Cpad;       // [3][Cpad] rotating
    float *red = smem + 4 * H + 6 * Cpad;             // [32]
    float *ldump = red + 64 + grp * H + n;            // lane-private sink for masked LDS ops

    const int g = (p.Bg == 1) ? 0 : b;
    const int Kin = p.Kin, Kout = p.Kout;
    const int32_t *il_in = p.il_in + (size_t)g * N * Kin;
    const float *w_in = p.w_in + (size_t)g * N * Kin;
    const float *term = p.term + (size_t)g * N;
    const float half_inf = p.neg_inf * 0.5f;
    const float NI2 = p.neg_inf * ASR_L2E;
    bool own = n < N;

    // ---- state-labelled check (lane n looks at state n) -------------------
    int label = 0;
    bool ok = true;
    if (own) {
        label = il_in[n * Kin];
        for (int k = 1; k < Kin; ++k)
            if (w_in[n * Kin + k] > half_inf && il_in[n * Kin + k] != label) ok = false;
        if (FL == 1 && (label < 0 || label >= C)) ok = false;
    }
    if (!__syncthreads_and(ok)) {
        lattice_fwbw_mitm_body<4, 8>(p, smem);
        return;
    }

    // ---- FL == 1: counting sort of the states by label ---------------------
    // lane n then holds the state at POSITION n; sid = its state id.
    int sid = n;
    float m1 = 0.f, m2 = 0.f, m4 = 0.f, m8 = 0.f, mb15 = 0.f, mb31 = 0.f;
    bool seg_plain = false, seg_multi = false;
    int *const srt = reinterpret_cast<int *>(red + 64 + 2 * H);
    int *const inv = srt + 2 * Cpad + 2 * H;          // [H] position -> state
    int *const pos = srt + 2 * Cpad;                  // [H] state -> position
    if (FL == 1) {
        int *const cnt = srt, *const cstart = srt + Cpad, *const slab = srt + 2 * Cpad + H;
        for (int c = threadIdx.x; c < Cpad; c += blockDim.x) cnt[c] = 0;
        for (int i = threadIdx.x; i < H; i += blockDim.x) slab[i] = 0x7fffffff;
        __syncthreads();
        int rank = 0;
        const int wv = threadIdx.x >> 6;
        for (int w = 0; w < (H >> 6); ++w) {          // the A group's waves in turn: a fixed order
            if (wv == w && own) rank = atomicAdd(&cnt[label], 1);
            __syncthreads();
        }
        if (threadIdx.x < (unsigned)C) {              // start of every class's run
            int acc = 0;
            for (int c = 0; c < (int)threadIdx.x; ++c) acc += cnt[c];
            cstart[threadIdx.x] = acc;
        }
        __syncthreads();
        if (grp == 0 && own) {
            const int pp = cstart[label] + rank;
            pos[n] = pp;
            inv[pp] = n;
            slab[pp] = label;
        }
        __syncthreads();
        // take over position n (positions 0..N-1 are exactly the N states)
        sid = own ? inv[n] : 0;
        const int sl = slab[n];                       // 0x7fffffff on empty positions
        label = own ? sl : 0;
        const int l = n & 63, wb = n & ~63, r16 = l >> 4;
        auto same = [&](int j) -> float { return (own && slab[j] == sl) ? 1.f : 0.f; };
        m1 = (l & 15) >= 1 ? same(n - 1) : 0.f;
        m2 = (l & 15) >= 2 ? same(n - 2) : 0.f;
        m4 = (l & 15) >= 4 ? same(n - 4) : 0.f;
        m8 = (l & 15) >= 8 ? same(n - 8) : 0.f;
        mb15 = (r16 & 1) ? same(wb + 16 * r16 - 1) : 0.f;
        mb31 = r16 >= 2 ? same(wb + 31) : 0.f;
        const bool seg_end = own && (l == 63 || slab[n + 1] != sl);
        const bool multi = own && (cstart[label] >> 6) != ((cstart[label] + cnt[label] - 1) >> 6);
        seg_plain = seg_end && !multi;
        seg_multi = seg_end && multi;
    }

    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const int m = len >> 1;                  // joint steps per phase
    const int solo = len - 2 * m;            // 1 if len is odd
    const int r = m % D, q = m / D;
    const size_t tstride = (size_t)p.B * C;
    float *grad_b = p.grad + (size_t)b * C;

    for (int t = len; t < p.T; ++t)                  // fst_utils.py:448
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            grad_b[(size_t)t * tstride + c] = 0.f;

    // ---- buffer resources: every global access of the scan goes through a
    // bounds-checked raw buffer with a per-lane 32-bit byte offset, so the
    // prefetch may run past either end of the utterance (out-of-range loads
    // return 0, out-of-range stores are dropped) and masked-off lanes simply
    // carry an out-of-range offset.  No clamps, no conditional VMEM: the
    // steady-state loop has an exact VMEM count per step and hipcc emits
    // counted vmcnt(N) waits.
    // workspace [T+2, B, H] (log2 units): slot t < m: alpha_{t+1};
    // slot t in [m, len]: beta_t.
    const u32 ts4 = (u32)tstride * 4u;               // frame stride of lp / grad, bytes
    const u32 as4 = (u32)p.B * (u32)H * 4u;          // slot stride of ws, bytes
    const u32 lp_bytes = (u32)(((size_t)p.T * p.B * C - (size_t)b * C) * 4);
    const u32 ws_bytes = (u32)(((size_t)(p.T + 2) * p.B * H - (size_t)b * H) * 4);
    const __amdgpu_buffer_rsrc_t lpR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.lp) + (size_t)b * C, 0, lp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gradR =
        __builtin_amdgcn_make_buffer_rsrc(grad_b, 0, lp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsR = __builtin_amdgcn_make_buffer_rsrc(
        p.alphas + (size_t)b * H, 0, ws_bytes, 0x00020000);
    const u32 OOB = 0xFFFFFFFFu;
    auto ld = [&](__amdgpu_buffer_rsrc_t R, u32 off) -> float {
#ifdef ASR_SL_NOMEM
        (void)R; return -1.f - 1e-9f * (float)off;
#else
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, off, 0, 0));
#endif
    };
    auto st = [&](__amdgpu_buffer_rsrc_t R, u32 off, float v) {
#ifdef ASR_SL_NOMEM
        (void)R; if (off == 12345u) red[0] = v;
#else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), R, off, 0, 0);
#endif
    };

    // ---- arcs of this group in registers --------------------------------
    const float *s0[K];
    float r_w[K];
    {
        const int Kg = grp ? Kout : Kin;
        const int32_t *oth = (grp ? p.dst_out : p.src_in) + (size_t)g * N * Kg;
        const float *wp = (grp ? p.w_out : p.w_in) + (size_t)g * N * Kg;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool v = own && k < Kg;
            const int o = v ? oth[sid * Kg + k] : 0;
            s0[k] = sbuf + (FL == 1 ? (v ? pos[o] : 0) : o);
            const float w = v ? wp[sid * Kg + k] : p.neg_inf;
            r_w[k] = w > half_inf ? w * ASR_L2E : NI2;
        }
    }
    float *const mine = sbuf + n;
    const float term2 = own ? fmaxf(term[sid], p.neg_inf) * ASR_L2E : NI2;
    float breg = term2;                               // B: beta_{t+1}[n] (log2)
    float logZ2 = 0.f;
    const int lane = threadIdx.x & 63;
    const int l0 = __builtin_amdgcn_readfirstlane(label);
    const bool shared_label = label == l0;
    const bool isB = grp != 0;                        // wave-uniform

    // per-group linear schedules (joint step i = 0..m-1 of each phase):
    //   phase 0: A frame i,   emission e_i,        store slot i
    //            B frame tb = len-1-solo-i, emission e_{tb-1}, store slot tb
    //   phase 1: A frame m+i, emission e_{m+i},    load slot m+i+1, flush frame m+i-1
    //            B frame m-1-i, emission e_{m-2-i}, load slot m-1-i, flush frame m-i
    const int dir = isB ? -1 : 1;
    const u32 estep = (u32)dir * ts4, wstep = (u32)dir * as4;
    const u32 lab4 = (u32)label * 4u, n4 = (u32)n * 4u;
    // workspace offsets of padding lanes (n >= N) start at 2^31: every buffer
    // is < 2^31 bytes (host check), so they stay out of range for any slot and
    // the padded columns cost no HBM traffic
    const u32 wn4 = own ? n4 : 0x80000000u;
    auto eoff = [&](int f) -> u32 { return lab4 + (u32)f * ts4; };
    auto woff = [&](int slot) -> u32 { return wn4 + (u32)slot * as4; };
    const int fE0 = isB ? len - 2 - solo : 0;
    const int sS0 = isB ? len - 1 - solo : 0;
    const int fE1 = isB ? m - 2 : m;
    const int sL1 = isB ? m - 1 : m + 1;
    const int fG1 = isB ? m + 1 : m - 2;      // frame of step i-2 at i = 0

    u32 wcur = 0;                     // phase 0: ws store offset of this step
    u32 gcur = 0;                     // phase 1: grad offset of the row to flush
    // phase 1 keeps three rotating gradient rows per group: the posteriors of
    // step i are added to row ra during step i+1 (while that step's LDS reads
    // are in flight) and the row is streamed out during step i+2.
    int ra = 0, rf = Cpad, rn = 2 * Cpad;
    float gprev = 0.f;                // posterior of the previous step, not yet added

    // side work of phase-1 step i: sum gprev per class into row ra, flush row rf
    // (frame of step i-2, offset gcur, masked off while i < 2), rotate.
    // FL == 1: segmented sum over the label-sorted lanes; the last lane of a run
    // stores the class total (one unconditional plain ds_write per lane, everything
    // else goes to a lane-private sink: an exact LDS op count keeps the compiler's
    // lgkmcnt waits counted); runs that cross a wave boundary add their pieces.
    // FL == 0: ONE unconditional ds_add per lane: lane 0 adds the DPP-reduced total
    // of the wave's shared label, lanes with another label add their own posterior,
    // zeros go to the sink (LDS float adds cost ~4 cycles per ACTIVE lane on gfx950;
    // the unconditional form was measured faster on the bigram numerator, 878 -> 783 us,
    // because it keeps the lgkmcnt waits counted).
    auto side_accumulate = [&]() {
        if (FL == 1) {
            // row_shr:1, 2, 4, 8, then row_bcast:15 -> rows 1, 3 and row_bcast:31 -> rows 2, 3
            const float v = seg_scan6(gprev, m1, m2, m4, m8, mb15, mb31);
            float *rw = row + ra + label;
            *(seg_plain ? rw : ldump) = v;
            if (seg_multi) atomicAdd(rw, v);
        } else {
            const float tot = dpp_wave_sum(shared_label ? gprev : 0.f);
            const float v = lane == 0 ? tot : (shared_label ? 0.f : gprev);
            float *dst = v != 0.f ? row + ra + label : ldump;
            atomicAdd(dst, v);
        }
    };
    auto side_rotate = [&]() {
        const int t = rf; rf = ra; ra = rn; rn = t;
    };

    // one recurrence step, identical instruction stream for both groups.
    // PH: phase; RD: LDS buffer read (writes RD^1); SOLO: 0 both groups
    // active, 1 only B, 2 only A (the idle group keeps its state).
    auto step = [&](auto ph_c, auto rd_c, auto solo_c, float ev, float wv, bool do_flush) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int RD = decltype(rd_c)::value;
        constexpr int SOLO = decltype(solo_c)::value;
        const int wr = (RD ^ 1) * H;
        const bool act = SOLO == 0 ? true : (SOLO == 1 ? isB : !isB);
        float x[K];
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = s0[k][RD * H];
        float fl = 0.f;
        const int ci = n < C ? n : 0;
        if (PH == 1 && FL == 1) fl = row[rf + ci];
        if (PH == 1) {
            // keep the LDS reads ahead of the DPP reduction: it runs in their shadow
            __builtin_amdgcn_sched_barrier(0);
            side_accumulate();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] += r_w[k];
        float mx, sum;
        if constexpr (K == 3 && FL == 1) {        // (C > H is bandwidth-bound: measured slower there)
            // the largest term contributes exp2(0) = 1: sort (max3 / med3 / min3) and take two
            // quarter-rate exponentials instead of three
            mx = __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), x[2]);
            const float md = __builtin_amdgcn_fmed3f(x[0], x[1], x[2]);
            const float mn = __builtin_fminf(__builtin_fminf(x[0], x[1]), x[2]);
            sum = 1.f + __builtin_amdgcn_exp2f(md - mx) + __builtin_amdgcn_exp2f(mn - mx);
        } else {
            mx = x[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
            sum = __builtin_amdgcn_exp2f(x[0] - mx);
#pragma unroll
            for (int k = 1; k < K; ++k) sum += __builtin_amdgcn_exp2f(x[k] - mx);
        }
        float val0 = mx + __builtin_amdgcn_logf(sum);     // B: beta_t
        float val1 = fmaf(ev, ASR_L2E, val0);             // A: alpha_{t+1}; B: beta_t + lp_{t-1}
        if (SOLO != 0) {
            const float keep = mine[RD * H];
            val1 = act ? val1 : keep;
        }
        mine[wr] = val1;
        if (PH == 0) {
            st(wsR, (SOLO == 0 || act) ? wcur : OOB, isB ? val0 : val1);
        } else {
            if (FL == 1) {
                const bool v = do_flush && n < C;
                *(v ? row + rf + ci : ldump) = 0.f;
                st(gradR, v ? gcur : OOB, fl);
            }
            float gam = __builtin_amdgcn_exp2f((isB ? breg : val1) + wv - logZ2) * p.gsign;
            if (!own || !act) gam = 0.f;
            gprev = gam;
            if (FL == 1) {
            } else if (do_flush) {
                for (int c = n; c < C; c += H) {
                    st(gradR, gcur + (u32)(c - n) * 4u, row[rf + c]);
                    row[rf + c] = 0.f;
                }
            }
            side_rotate();
        }
        if (SOLO == 0 || act) breg = val0;
        SL_BARRIER();
    };
    // phase-1 drain step: side work only
    auto drain = [&](bool do_flush) {
        const int ci = n < C ? n : 0;
        side_accumulate();
        gprev = 0.f;
        if (FL == 1) {
            const float fl = row[rf + ci];
            const bool v = do_flush && n < C;
            *(v ? row + rf + ci : ldump) = 0.f;
            st(gradR, v ? gcur : OOB, fl);
        } else if (do_flush) {
            for (int c = n; c < C; c += H) {
                st(gradR, gcur + (u32)(c - n) * 4u, row[rf + c]);
                row[rf + c] = 0.f;
            }
        }
        side_rotate();
        __syncthreads();
    };
    auto copy_back = [&]() {          // buffer 1 -> buffer 0 (restore parity)
        mine[0] = mine[H];
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- initial state ----------------------------------------------------
    {
        const float e_last = ld(lpR, eoff(len - 1));          // OOB -> 0 when len == 0
        mine[0] = isB ? (own ? fmaf(e_last, ASR_L2E, term2) : NI2)   // beta_len + lp_{len-1}
                      : ((own && sid == 0) ? 0.f : NI2);             // alpha_0
        st(wsR, isB ? woff(len) : OOB, term2);                // slot len = beta_len
    }
    for (int c = n; c < 3 * Cpad; c += H) row[c] = 0.f;
    __syncthreads();

    float e[D], eh[D], wsr[D], wh[D];
#ifdef ASR_SL_STAMPS
    unsigned long long stamp[6];
#define SL_STAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define SL_STAMP(i) do {} while (0)
#endif
    SL_STAMP(0);

    // ================= phase 0 ============================================
    if (solo) {                                   // B alone: frame len-1
        const float ev = ld(lpR, isB ? eoff(len - 2) : OOB);
        wcur = woff(len - 1);
        step(I0(), I0(), I1(), ev, 0.f, false);
        copy_back();
    }
#pragma unroll
    for (int u = 0; u < D; ++u) {
        eh[u] = ld(lpR, eoff(fE0 + dir * u));
        e[u] = ld(lpR, eoff(fE0 + dir * (r + u)));
    }
    wcur = woff(sS0);
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (u < r) {
            if (u & 1) step(I0(), I1(), I0(), eh[u], 0.f, false);
            else step(I0(), I0(), I0(), eh[u], 0.f, false);
            wcur += wstep;
        }
    if (r & 1) copy_back();
    {
        u32 epf = eoff(fE0 + dir * (r + D));      // emission of step i + D
        for (int c = 0; c < q; ++c) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (u & 1) step(I0(), I1(), I0(), e[u], 0.f, false);
                else step(I0(), I0(), I0(), e[u], 0.f, false);
                wcur += wstep;
                e[u] = ld(lpR, epf);
                epf += estep;
            }
        }
    }

    SL_STAMP(1);
    // ================= meeting point ======================================
#pragma unroll
    for (int u = 0; u < D; ++u) {                 // prime phase-1 rings early
        eh[u] = ld(lpR, eoff(fE1 + dir * u));
        e[u] = ld(lpR, eoff(fE1 + dir * (r + u)));
        wh[u] = ld(wsR, woff(sL1 + dir * u));
        wsr[u] = ld(wsR, woff(sL1 + dir * (r + u)));
    }
    if (isB) mine[H] = breg;                      // pure beta_m in the spare buffer
    __syncthreads();
    {
        const float *al = smem;                   // alpha_m: A buffer 0
        const float *be = smem + 3 * H;           // beta_m:  B buffer 1
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < N; i += blockDim.x) mx = fmaxf(mx, al[i] + be[i]);
        mx = block_max(mx, red);
        float sum = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            sum += __builtin_amdgcn_exp2f(al[i] + be[i] - mx);
        sum = block_sum(sum, red);
        logZ2 = mx + __builtin_amdgcn_logf(sum);
        // = logsumexp_n(alpha_len + terminal) (fst_utils.py:445): the total over all paths
        // is the same at every frame; taking it here saves a block reduction at the end
        if (threadIdx.x == 0) p.logZ[b] = logZ2 * ASR_LN2;
    }
    __syncthreads();

    SL_STAMP(2);
    // ================= phase 1 ============================================
    const u32 gstep = estep;
    gcur = n4 + (u32)fG1 * ts4;                   // row flushed at step 0 (masked off)
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (u < r) {
            if (u & 1) step(I1(), I1(), I0(), eh[u], wh[u], u > 1);
            else step(I1(), I0(), I0(), eh[u], wh[u], u > 1);
            gcur += gstep;
        }
    if (r & 1) copy_back();
    {
        u32 epf = eoff(fE1 + dir * (r + D));
        u32 wpf = woff(sL1 + dir * (r + D));
        for (int c = 0; c < q; ++c) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const bool fl = (r + c * D + u) > 1;
                if (u & 1) step(I1(), I1(), I0(), e[u], wsr[u], fl);
                else step(I1(), I0(), I0(), e[u], wsr[u], fl);
                gcur += gstep;
                e[u] = ld(lpR, epf);
                wsr[u] = ld(wsR, wpf);
                epf += estep;
                wpf += wstep;
            }
        }
    }
    SL_STAMP(3);
    if (solo) {                                   // A alone: frame len-1
        const float ev = ld(lpR, isB ? OOB : eoff(len - 1));
        const float wv = ld(wsR, isB ? OOB : woff(len));       // beta_len
        step(I1(), I0(), I2(), ev, wv, m > 1);
        gcur += gstep;
    }
    {   // drain: rows of the last two steps (B's row for A's solo step is
        // empty and its frame offset is out of range)
        const int S = m + solo;
        drain(S > 1);
        gcur += gstep;
        drain(S > 0);
    }
    // fst_utils.py:476: logsumexp(alpha_0 + beta_0) == beta_0[0]
    if (p.logZ_bwd && isB && own && sid == 0) p.logZ_bwd[b] = breg * ASR_LN2;
#ifdef ASR_SL_STAMPS
    SL_STAMP(4);
    __syncthreads();
    if (threadIdx.x == 0 && p.T > 0) {
        float *o = grad_b + (size_t)(p.T - 1) * tstride;
        for (int i = 0; i < 4; ++i) o[i] = (float)(stamp[i + 1] - stamp[i]);
    }
#endif
}

#include "lattice_band.inc"

struct FwdParams {
    const float *lp;
    int T, B, C;
    const int32_t *lens;
    const int32_t *src_in, *il_in;
    const float *w_in, *term;
    int N, K, Bg;
    float neg_inf;
    float *score;
    int32_t *best_il;
    uint16_t *bp;  // [T,B,N] arg-max arc slot, viterbi only
};

template <bool VITERBI>
__global__ void lattice_forward_kernel(FwdParams p) {
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int N = p.N, C = p.C, K = p.K;
    const int Npad = (N + 3) & ~3;
    float *abuf = smem;                 // [2][Npad]
    float *red = smem + 2 * Npad;       // [64]
    int *redi = (int *)(red + 32);

    const int g = (p.Bg == 1) ? 0 : b;
    const int32_t *src_in = p.src_in + (size_t)g * N * K;
    const int32_t *il_in = p.il_in + (size_t)g * N * K;
    const float *w_in = p.w_in + (size_t)g * N * K;
    const float *term = p.term + (size_t)g * N;
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;
    const float *lp_b = p.lp + (size_t)b * C;
    const bool want_path = VITERBI && p.best_il != nullptr;
    uint16_t *bp_b = want_path ? p.bp + (size_t)b * N : nullptr;
    const size_t bstride = (size_t)p.B * N;

    for (int n = tid; n < Npad; n += NT) abuf[n] = (n == 0) ? 0.f : p.neg_inf;
    if (want_path)
        for (int t = len + tid; t < p.T; t += NT) p.best_il[(size_t)t * p.B + b] = 0;
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < len; ++t) {
        const float *a = abuf + cur * Npad;
        float *an = abuf + (cur ^ 1) * Npad;
        const float *lrow = lp_b + (size_t)t * tstride;
        for (int n = tid; n < N; n += NT) {
            if (VITERBI) {
                float best = -INFINITY;
                int arg = 0;
                for (int k = 0; k < K; ++k) {
                    int i = n * K + k;
                    // same association as the reference: (alpha + w) + lp (:387-390)
                    float v = (a[src_in[i]] + w_in[i]) + lrow[il_in[i]];
                    if (v > best) { best = v; arg = k; }
                }
                an[n] = best;
                if (want_path) bp_b[(size_t)t * bstride + n] = (uint16_t)arg;
            } else {
                Lse acc;
                acc.init();
                for (int k = 0; k < K; ++k) {
                    int i = n * K + k;
                    acc.add((a[src_in[i]] + w_in[i]) + lrow[il_in[i]]);
                }
                an[n] = acc.value();
            }
        }
        cur ^= 1;
        __syncthreads();
    }

    const float *a = abuf + cur * Npad;
    if (VITERBI) {
        // first maximum over n of alpha + terminal (:396)
        float best = -INFINITY;
        int arg = 0x7fffffff;
        for (int n = tid; n < N; n += NT) {
            float v = a[n] + term[n];
            if (v > best) { best = v; arg = n; }
        }
        // wave arg-max with lowest-index tie-break
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        const int lane = tid & 63, w = tid >> 6, nw = (NT + 63) >> 6;
        if (lane == 0) { red[w] = best; redi[w] = arg; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < nw; ++i)
                if (red[i] > best || (red[i] == best && redi[i] < arg)) {
                    best = red[i];
                    arg = redi[i];
                }
            p.score[b] = best;
            if (want_path) {
                int st = arg;
                for (int t = len - 1; t >= 0; --t) {
                    int k = bp_b[(size_t)t * bstride + st];
                    p.best_il[(size_t)t * p.B + b] = il_in[st * K + k];
                    st = src_in[st * K + k];
                }
            }
        }
    } else {
        float m = -INFINITY;
        for (int n = tid; n < N; n += NT) m = fmaxf(m, a[n] + term[n]);
        m = block_max(m, red);
        float s = 0.f;
        for (int n = tid; n < N; n += NT) s += __expf(a[n] + term[n] - m);
        s = block_sum(s, red);
        if (tid == 0) p.score[b] = m + __logf(s);
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int64_t asr_lattice_fwbw_workspace_bytes(int T, int B, int C, int N) {
    (void)C;
    if (T < 0 || B < 0 || N < 0) return -1;
    // [T+2, B, round_up(N,64)] f32 (+1 slot: beta_len, +1: spare); the band kernel keeps one
    // exponent word per lane and row beside its value rows: + [T+2, B, 64]
    const int64_t H = (N + 63) / 64 * 64;
    return (int64_t)(T + 2) * B * (H + 64) * (int64_t)sizeof(float) + 256;
}

extern "C" int asr_lattice_fwbw_signed_f32(const float *lp, int T, int B, int C,
                                    const int32_t *lens,
                                    const int32_t *src_in, const int32_t *il_in,
                                    const float *w_in, const float *term,
                                    const int32_t *dst_out, const int32_t *il_out,
                                    const float *w_out,
                                    int N, int Kin, int Kout, int Bg, float neg_inf,
                                    float grad_sign,
                                    float *out_logZ, float *out_grad,
                                    float *out_logZ_bwd,
                                    void *workspace, int64_t workspace_bytes,
                                    void *stream) {
    if (!(grad_sign == 1.f || grad_sign == -1.f)) return ASR_EINVAL;
    if (T < 0 || B < 0 || C <= 0 || N <= 0 || Kin <= 0 || Kout <= 0) return ASR_EINVAL;
    if (Bg != 1 && Bg != B) return ASR_EINVAL;            // fst_utils.py:406
    if (B == 0) return ASR_OK;
    if (!lp && T > 0) return ASR_EINVAL;
    if (!lens || !src_in || !il_in || !w_in || !term || !dst_out || !il_out ||
        !w_out || !out_logZ || (!out_grad && T > 0))
        return ASR_EINVAL;
    if (workspace_bytes < asr_lattice_fwbw_workspace_bytes(T, B, C, N) ||
        (!workspace && T > 0))
        return ASR_EINVAL;
    if (!(neg_inf < 0.f)) return ASR_EINVAL;

    FwbwParams p;
    p.lp = lp; p.T = T; p.B = B; p.C = C; p.lens = lens;
    p.src_in = src_in; p.il_in = il_in; p.w_in = w_in; p.term = term;
    p.dst_out = dst_out; p.il_out = il_out; p.w_out = w_out;
    p.N = N; p.Kin = Kin; p.Kout = Kout; p.Bg = Bg; p.neg_inf = neg_inf;
    p.logZ = out_logZ; p.grad = out_grad; p.logZ_bwd = out_logZ_bwd;
    p.alphas = (float *)workspace;
    p.redo = nullptr;
    p.gsign = grad_sign;
    p.ctc_labels = p.ctc_label_lens = nullptr; p.ctc_lmax = 0;

    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    size_t lds = (size_t)(2 * Npad + 2 * Cpad + 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int Kmax = Kin > Kout ? Kin : Kout;
    void (*kern)(FwbwParams);
    int nt;
    const size_t lds_mitm = (size_t)(4 * Npad + 4 * Cpad + 64) * sizeof(float);
    // the scan addresses lp / grad through raw buffers with UNSIGNED 32-bit byte offsets
    // (modular arithmetic, bounds-checked by the hardware against num_records): the tensor and
    // the frames the prefetch may run past either end (+- 64 covers D + the half-step overrun)
    // must stay below 4 GiB so that an overrun offset cannot wrap into the tensor
    const bool fits32 = (size_t)(T + 64) * B * C * 4 < (1ull << 32) &&
                        (size_t)(T + 2) * B * round_up(N, 64) * 4 < (1ull << 30);
    if (N <= 512 && Kmax <= 4 && lds_mitm <= 160 * 1024 && fits32) {
        // state-labelled fast path; a workgroup whose graph fails the entry
        // check runs the generic body inside the same launch
        const int H = round_up(N, 64);
        // + the label sort of FL == 1: cnt, cstart [Cpad]; pos, slab, inv [H]
        const size_t lds_sl = (size_t)(4 * H + 6 * Cpad + 64 + 2 * H + 2 * Cpad + 3 * H) * sizeof(float);
        if (C <= H)
            kern = Kmax <= 3 ? lattice_fwbw_sl_kernel<3, 8, 1> : lattice_fwbw_sl_kernel<4, 8, 1>;
        else
            kern = Kmax <= 3 ? lattice_fwbw_sl_kernel<3, 8, 0> : lattice_fwbw_sl_kernel<4, 8, 0>;
        nt = 2 * H;
        lds = lds_sl > lds_mitm ? lds_sl : lds_mitm;
    } else if (N <= 1024 && Kmax <= 4) {
        kern = lattice_fwbw_kernel<4>;
        nt = round_up(N, 64);
    } else {
        kern = lattice_fwbw_kernel<0>;
        nt = N >= 1024 ? 1024 : round_up(N, 64);
    }
    if (lds > 160 * 1024) return ASR_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return ASR_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_lattice_fwbw_f32(const float *lp, int T, int B, int C,
                                    const int32_t *lens,
                                    const int32_t *src_in, const int32_t *il_in,
                                    const float *w_in, const float *term,
                                    const int32_t *dst_out, const int32_t *il_out,
                                    const float *w_out,
                                    int N, int Kin, int Kout, int Bg, float neg_inf,
                                    float *out_logZ, float *out_grad,
                                    float *out_logZ_bwd,
                                    void *workspace, int64_t workspace_bytes,
                                    void *stream) {
    return asr_lattice_fwbw_signed_f32(lp, T, B, C, lens, src_in, il_in, w_in, term, dst_out, il_out, w_out,
                                       N, Kin, Kout, Bg, neg_inf, 1.f, out_logZ, out_grad, out_logZ_bwd,
                                       workspace, workspace_bytes, stream);
}

extern "C" int64_t asr_lattice_viterbi_workspace_bytes(int T, int B, int N) {
    if (T < 0 || B < 0 || N < 0) return -1;
    return (int64_t)T * B * N * (int64_t)sizeof(uint16_t) + 256;
}

extern "C" int asr_lattice_forward_f32(const float *lp, int T, int B, int C,
                                       const int32_t *lens,
                                       const int32_t *src_in, const int32_t *il_in,
                                       const float *w_in, const float *term,
                                       int N, int K, int Bg, float neg_inf,
                                       int viterbi,
                                       float *out_score, int32_t *out_best_il,
                                       void *workspace, int64_t workspace_bytes,
                                       void *stream) {
    if (T < 0 || B < 0 || C <= 0 || N <= 0 || K <= 0 || K > 65535) return ASR_EINVAL;
    if (Bg != 1 && Bg != B) return ASR_EINVAL;            // fst_utils.py:350
    if (B == 0) return ASR_OK;
    if (!lp && T > 0) return ASR_EINVAL;
    if (!lens || !src_in || !il_in || !w_in || !term || !out_score) return ASR_EINVAL;
    if (!(neg_inf < 0.f)) return ASR_EINVAL;
    const bool want_path = viterbi && out_best_il;
    if (want_path && T > 0 &&
        (!workspace || workspace_bytes < asr_lattice_viterbi_workspace_bytes(T, B, N)))
        return ASR_EINVAL;

    FwdParams p;
    p.lp = lp; p.T = T; p.B = B; p.C = C; p.lens = lens;
    p.src_in = src_in; p.il_in = il_in; p.w_in = w_in; p.term = term;
    p.N = N; p.K = K; p.Bg = Bg; p.neg_inf = neg_inf;
    p.score = out_score;
    p.best_il = want_path ? out_best_il : nullptr;
    p.bp = (uint16_t *)workspace;

    const int Npad = (N + 3) & ~3;
    const size_t lds = (size_t)(2 * Npad + 64) * sizeof(float);
    if (lds > 160 * 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    void (*kern)(FwdParams) =
        viterbi ? lattice_forward_kernel<true> : lattice_forward_kernel<false>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return ASR_EUNSUPPORTED;
    }
    const int nt = N >= 1024 ? 1024 : round_up(N, 64);
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// Band lattices (CTC numerators of mono-character transcripts) in the rescaled linear domain:
// lattice_band.inc.  Same arguments and outputs as asr_lattice_fwbw_f32; the caller asserts
// nothing — every utterance's graph is checked in the kernel and anything else runs the
// generic log-domain body in the same launch (correct, slow), so callers route here only
// graphs they know (or have checked on the host) to be band-shaped.
extern "C" int asr_lattice_fwbw_band_supported(int T, int B, int C, int N, int Kin, int Kout, int Bg) {
    const int Kmax = Kin > Kout ? Kin : Kout;
    return T > 0 && B > 0 && C > 0 && C <= 64 && N > 0 && N <= band::NS && Kmax <= 4 && Bg == B &&
           (size_t)T * B * C * 4 < (1ull << 31) &&
           (size_t)(T + 2) * (round_up(N, 64) + 64) * 4 < (1ull << 31);
}

extern "C" int asr_lattice_fwbw_band_f32(const float *lp, int T, int B, int C,
                                         const int32_t *lens,
                                         const int32_t *src_in, const int32_t *il_in,
                                         const float *w_in, const float *term,
                                         const int32_t *dst_out, const int32_t *il_out,
                                         const float *w_out,
                                         int N, int Kin, int Kout, int Bg, float neg_inf, float grad_sign,
                                         float *out_logZ, float *out_grad,
                                         float *out_logZ_bwd,
                                         void *workspace, int64_t workspace_bytes, uint32_t *redo_count,
                                         const int32_t *ctc_labels, const int32_t *ctc_label_lens, int ctc_lmax,
                                         void *stream) {
    if (T < 0 || B < 0 || C <= 0 || N <= 0 || Kin <= 0 || Kout <= 0) return ASR_EINVAL;
    if (Bg != 1 && Bg != B) return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!lp || !lens || !src_in || !il_in || !w_in || !term || !dst_out || !il_out ||
        !w_out || !out_logZ || !out_grad || !workspace)
        return ASR_EINVAL;
    if (workspace_bytes < asr_lattice_fwbw_workspace_bytes(T, B, C, N)) return ASR_EINVAL;
    if (!(neg_inf < 0.f)) return ASR_EINVAL;
    if (!asr_lattice_fwbw_band_supported(T, B, C, N, Kin, Kout, Bg)) return ASR_EUNSUPPORTED;

    FwbwParams p;
    p.lp = lp; p.T = T; p.B = B; p.C = C; p.lens = lens;
    p.src_in = src_in; p.il_in = il_in; p.w_in = w_in; p.term = term;
    p.dst_out = dst_out; p.il_out = il_out; p.w_out = w_out;
    p.N = N; p.Kin = Kin; p.Kout = Kout; p.Bg = Bg; p.neg_inf = neg_inf;
    p.logZ = out_logZ; p.grad = out_grad; p.logZ_bwd = out_logZ_bwd;
    p.alphas = (float *)workspace;
    p.redo = nullptr;
    if (!(grad_sign == 1.f || grad_sign == -1.f)) return ASR_EINVAL;
    p.gsign = grad_sign;
    if (ctc_labels && (!ctc_label_lens || ctc_lmax < 0 || 2 * ctc_lmax + 1 > N)) return ASR_EINVAL;
    p.ctc_labels = ctc_labels; p.ctc_label_lens = ctc_labels ? ctc_label_lens : nullptr; p.ctc_lmax = ctc_lmax;
    // the in-kernel fallback (lattice_fwbw_generic_body<0>) needs 2 Npad + 2 Cpad + 64 words
    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    size_t lds = (size_t)band::LDS_WORDS * sizeof(float);
    const size_t lds_gen = (size_t)(2 * Npad + 2 * Cpad + 64) * sizeof(float);
    if (lds_gen > lds) lds = lds_gen;
    const int grid = 8 * ((B + 7) / 8);       // blocks b, b + 8, ... (one XCD) take neighbouring utterances
    // the count of utterances redone by the fallback body: a RUNNING counter of the caller's
    // (never reset here: a memset in front of every launch cost more than 5 % of the launch)
    p.redo = redo_count;
    hipLaunchKernelGGL(lattice_fwbw_band_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
