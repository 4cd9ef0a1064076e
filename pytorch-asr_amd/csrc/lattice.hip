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
    float *alphas;  // [T,B,N]
    int *skip;      // [B] or null: utterances already done by the band kernel
    int *split;     // [B] (FL == 2): 1 = posteriors left in ws for lattice_scatter_kernel
};

// KR > 0: every thread owns ONE state (N <= blockDim) and keeps its <= KR
// in-arcs / out-arcs in registers for the whole scan.  KR == 0: states are
// strided over the block and arcs are streamed from global memory (L2) every
// frame (large shared graphs such as the CTC-G denominator).
template <int KR>
__global__ void lattice_fwbw_kernel(FwbwParams p) {
    extern __shared__ float smem[];
    const int b = blockIdx.x;
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
    const size_t astride = (size_t)p.B * N;
    float *alphas_b = p.alphas + (size_t)b * N;
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
                        float o = __expf(v[k] + a);
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
                        float o = __expf(v + a);
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
                                float o = __expf(v[k] + a);
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
// i.e. ONE emission fetch, ONE posterior exp and ONE LDS add per state and
// frame instead of one per arc.  Same meet-in-the-middle schedule as
// lattice_fwbw_mitm_kernel.  All scores are kept in log2 units so the
// recurrences use the raw v_exp_f32 / v_log_f32 (no range-reduction code);
// results are converted back on output.  The label shared by most lanes of a
// wave (the blank, label of lane 0's state) is reduced with DPP before the LDS
// add so it costs one ds_add per wave instead of a 32-way same-address add.
//
// ws[t,b,n] (log2 units): alpha_{t+1}[n] for t < m, beta_{t+1}[n] for t >= m.
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
    // FL == 1: C <= H, the per-step row flush is one store per lane;
    // FL == 0: runtime flush loop (large C, bandwidth-bound regime);
    // FL == 2: split scatter — phase 1 leaves the state posteriors gamma_f[n] in the
    //          workspace (in place of the value it consumed: slot f for f < m, slot f+1
    //          for f >= m) and lattice_scatter_kernel sums them per class afterwards at
    //          full occupancy, so nothing but the recurrence sits on the sequential chain.
    extern __shared__ float smem[];
    typedef unsigned int u32;
    const int b = blockIdx.x;
    if (p.skip && p.skip[b]) {                        // done by lattice_fwbw_band_kernel
        if (FL == 2 && threadIdx.x == 0) p.split[b] = 0;
        return;
    }
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
    const bool own = n < N;

    // ---- state-labelled check -------------------------------------------
    int label = 0;
    bool ok = true;
    if (own) {
        label = il_in[n * Kin];
        for (int k = 1; k < Kin; ++k)
            if (w_in[n * Kin + k] > half_inf && il_in[n * Kin + k] != label) ok = false;
    }
    if (!__syncthreads_and(ok)) {
        if (FL == 2 && threadIdx.x == 0) p.split[b] = 0;
        lattice_fwbw_mitm_body<4, 8>(p, smem);
        return;
    }
    if (FL == 2 && threadIdx.x == 0) p.split[b] = 1;

    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const int m = len >> 1;                  // joint steps per phase
    const int solo = len - 2 * m;            // 1 if len is odd
    const int r = m % D, q = m / D;
    const size_t tstride = (size_t)p.B * C;
    float *grad_b = p.grad + (size_t)b * C;

    if (FL != 2)
        for (int t = len; t < p.T; ++t)              // fst_utils.py:448
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
            s0[k] = sbuf + (v ? oth[n * Kg + k] : 0);
            const float w = v ? wp[n * Kg + k] : p.neg_inf;
            r_w[k] = w > half_inf ? w * ASR_L2E : NI2;
        }
    }
    float *const mine = sbuf + n;
    const float term2 = own ? fmaxf(term[n], p.neg_inf) * ASR_L2E : NI2;
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

    // side work of phase-1 step i: add gprev into row ra, flush row rf
    // (frame of step i-2, offset gcur, masked off while i < 2), rotate.
    // ONE unconditional ds_add per lane (exact LDS op count keeps the
    // compiler's lgkmcnt waits counted): lane 0 adds the DPP-reduced total of
    // the wave's shared label, lanes with another label add their own
    // posterior, everything else (zeros) goes to a lane-private sink.
    // (LDS float adds cost ~4 cycles per ACTIVE lane on gfx950: for small C
    // only lanes with a non-zero, non-shared posterior issue one; for large C
    // the single unconditional form is faster — measured 878 -> 783 us on the
    // bigram numerator — because it keeps the lgkmcnt waits counted.)
    auto side_accumulate = [&]() {
        const float tot = dpp_wave_sum(shared_label ? gprev : 0.f);
        if (FL == 1) {
            float *rw = row + ra;
            if (lane == 0 && tot != 0.f) atomicAdd(&rw[l0], tot);
            if (!shared_label && gprev != 0.f) atomicAdd(&rw[label], gprev);
        } else {
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
        if (PH == 1 && FL != 2) {
            // keep the LDS reads ahead of the DPP reduction: it runs in their shadow
            __builtin_amdgcn_sched_barrier(0);
            side_accumulate();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] += r_w[k];
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) sum += __builtin_amdgcn_exp2f(x[k] - mx);
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
            float gam = __builtin_amdgcn_exp2f((isB ? breg : val1) + wv - logZ2);
            if (!own || !act) gam = 0.f;
            gprev = gam;
            if (FL == 2) {
                st(wsR, (SOLO == 0 || act) ? gcur : OOB, gam);
            } else if (FL == 1) {
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
        if (FL == 2) return;
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
                      : ((n == 0) ? 0.f : NI2);                      // alpha_0
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
    }
    __syncthreads();

    SL_STAMP(2);
    // ================= phase 1 ============================================
    // FL == 2: gcur walks the workspace slots this group loads (gamma goes back in place)
    const u32 gstep = FL == 2 ? wstep : estep;
    gcur = FL == 2 ? woff(sL1) : n4 + (u32)fG1 * ts4;   // FL != 2: row flushed at step 0 (masked off)
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
    {   // logZ = logsumexp_n(alpha_len + terminal) (fst_utils.py:445)
        const float *al = smem + (solo ? H : 0);
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            mx = fmaxf(mx, al[i] + fmaxf(term[i], p.neg_inf) * ASR_L2E);
        mx = block_max(mx, red);
        float sum = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            sum += __builtin_amdgcn_exp2f(al[i] + fmaxf(term[i], p.neg_inf) * ASR_L2E - mx);
        sum = block_sum(sum, red);
        if (threadIdx.x == 0) p.logZ[b] = (mx + __builtin_amdgcn_logf(sum)) * ASR_LN2;
    }
    // fst_utils.py:476: logsumexp(alpha_0 + beta_0) == beta_0[0]
    if (p.logZ_bwd && isB && n == 0) p.logZ_bwd[b] = breg * ASR_LN2;
#ifdef ASR_SL_STAMPS
    SL_STAMP(4);
    __syncthreads();
    if (threadIdx.x == 0 && p.T > 0) {
        float *o = grad_b + (size_t)(p.T - 1) * tstride;
        for (int i = 0; i < 4; ++i) o[i] = (float)(stamp[i + 1] - stamp[i]);
    }
#endif
}

// ---------------------------------------------------------------------------
// Second half of the split scatter (lattice_fwbw_sl_kernel<.., 2>): per frame and
// utterance, grad[f,b,c] = sum_{n: label n = c} gamma_f[n] with gamma read from the
// workspace (slot f for f < len/2, slot f+1 above), zeros for f >= len
// (fst_utils.py:448).  One 4-wave workgroup per utterance and chunk of 4*FPW
// frames; a wave takes one frame per iteration: H/64 coalesced loads per lane,
// the label of state 0 (the blank of a CTC chain, shared by half the states) is
// summed with DPP, the others with LDS float adds into the wave's own bins.
// ---------------------------------------------------------------------------
template <int NJ, int FPW>
__global__ __launch_bounds__(256) void lattice_scatter_kernel(FwbwParams p) {
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    if (!p.split[b]) return;
    const int N = p.N, C = p.C, H = NJ * 64;
    const int Cpad = (C + 3) & ~3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *bins = smem + wave * Cpad;
    const int g = (p.Bg == 1) ? 0 : b;
    const int32_t *il_in = p.il_in + (size_t)g * N * p.Kin;
    int lab[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = lane + 64 * j;
        lab[j] = n < N ? il_in[(size_t)n * p.Kin] : -1;
    }
    const int l0 = __builtin_amdgcn_readfirstlane(lab[0]);
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const int m = len >> 1;
    for (int c = lane; c < Cpad; c += 64) bins[c] = 0.f;
    const int f0 = blockIdx.y * (4 * FPW) + wave;
    float v[FPW][NJ];
#pragma unroll
    for (int it = 0; it < FPW; ++it) {
        const int f = f0 + 4 * it;
        const int slot = f < m ? f : f + 1;
        const float *src = p.alphas + ((size_t)slot * p.B + b) * H;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            v[it][j] = (f < len && lab[j] >= 0) ? src[lane + 64 * j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < FPW; ++it) {
        const int f = f0 + 4 * it;
        float t0 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool sh = lab[j] == l0;
            t0 += sh ? v[it][j] : 0.f;
            if (!sh && v[it][j] != 0.f) atomicAdd(&bins[lab[j]], v[it][j]);
        }
        const float tot = dpp_wave_sum(t0);
        __syncthreads();
        float *dst = p.grad + ((size_t)f * p.B + b) * C;
        for (int c = lane; c < C; c += 64) {
            const float x = bins[c] + (c == l0 ? tot : 0.f);
            bins[c] = 0.f;
            if (f < p.T) dst[c] = x;
        }
        __syncthreads();
    }
}

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

// ---------------------------------------------------------------------------
// BAND lattices: state-labelled graphs whose in-arcs come from states
// {n, n-1, n-2} and whose out-arcs go to {n, n+1, n+2} — every CTC numerator
// lattice of the reference (compose(decoding_fst, chain), fst_utils.py:603-613:
// blank_0, label_0, blank_1, ... with self loops, next-state arcs and the
// skip-the-blank arc).  For these the scan needs no LDS exchange at all:
//   * ONE wave per chain: wave 0 runs alpha, wave 1 runs beta (meet in the
//     middle as above); a lane keeps S consecutive states in registers and gets
//     the two neighbouring states of the next/previous lane with DPP wave
//     shifts, so there is no barrier and no LDS round trip on the recurrence;
//   * the log-prob row of a frame is loaded ONCE per chain (lane c holds class
//     c, C <= 64) D frames ahead and the per-state emissions are gathered from
//     it with ds_bpermute;
//   * posteriors are summed per class without atomics: the slots whose states
//     all carry one label (the blanks) are reduced with one DPP sum, every
//     other class sums its <= 16 states through a gather list built once per
//     utterance.
// The two waves only meet at the phase boundary (alpha_m / beta_m exchange ->
// logZ).  Graphs that are not bands (or N > 64*S, C > 64, a class with more
// than 16 states) leave skip[b] = 0 and are handled by the kernels above.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_shr1(float v, float fill) {     // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {     // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
// log2(2^x0 + 2^x1 + 2^x2): the largest term contributes exactly 1, so two
// v_exp_f32 (quarter rate) instead of three, for three full-rate min/med/max
__device__ __forceinline__ float lse3_2(float x0, float x1, float x2) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(x0, x1), x2);
    const float lo = __builtin_fminf(__builtin_fminf(x0, x1), x2);
    const float mid = __builtin_amdgcn_fmed3f(x0, x1, x2);
    const float s = 1.f + __builtin_amdgcn_exp2f(mid - m) + __builtin_amdgcn_exp2f(lo - m);
    return m + __builtin_amdgcn_logf(s);
}

#define BAND_LIST 16

template <int S, int D>
__global__ __launch_bounds__(128) void lattice_fwbw_band_kernel(FwbwParams p) {
    typedef unsigned int u32;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    static_assert(S == 4, "ws rows are moved as one 16-byte access per lane");
    __shared__ float xch[2][64 * S];                   // alpha_m | beta_m
    __shared__ __attribute__((aligned(16))) float gbuf[2][64 * S + 4];   // posteriors per chain, [64*S] stays 0
    __shared__ int labtab[64 * S];
    __shared__ unsigned short lists[64][2][BAND_LIST];
    __shared__ int lcnt[64][2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool isB = __builtin_amdgcn_readfirstlane(tid >> 6) != 0;
    const int N = p.N, C = p.C;
    const int g = (p.Bg == 1) ? 0 : b;
    const int Kin = p.Kin, Kout = p.Kout;
    const int32_t *src_in = p.src_in + (size_t)g * N * Kin;
    const int32_t *il_in = p.il_in + (size_t)g * N * Kin;
    const float *w_in = p.w_in + (size_t)g * N * Kin;
    const int32_t *dst_out = p.dst_out + (size_t)g * N * Kout;
    const float *w_out = p.w_out + (size_t)g * N * Kout;
    const float *term = p.term + (size_t)g * N;
    const float half_inf = p.neg_inf * 0.5f;
    const float NI2 = p.neg_inf * ASR_L2E;

    // ---- per-lane states n = S*lane + s: band weights, labels, structure check
    int lab[S];
    float wb[S][3], term2[S];
    bool dead[S];
    bool ok = true;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = lane * S + s;
        const bool valid = n < N;
        lab[s] = valid ? il_in[(size_t)n * Kin] : 0;
        if (valid && (lab[s] < 0 || lab[s] >= C)) ok = false;
        float wi0 = NI2, wi1 = NI2, wi2 = NI2, wo0 = NI2, wo1 = NI2, wo2 = NI2;
        if (valid) {
            for (int k = 0; k < Kin; ++k) {
                const float w = w_in[(size_t)n * Kin + k];
                if (w > half_inf) {
                    const int d = n - src_in[(size_t)n * Kin + k];
                    const float w2 = w * ASR_L2E;
                    if (il_in[(size_t)n * Kin + k] != lab[s]) ok = false;
                    if (d == 0 && wi0 == NI2) wi0 = w2;
                    else if (d == 1 && wi1 == NI2) wi1 = w2;
                    else if (d == 2 && wi2 == NI2) wi2 = w2;
                    else ok = false;
                }
            }
            for (int k = 0; k < Kout; ++k) {
                const float w = w_out[(size_t)n * Kout + k];
                if (w > half_inf) {
                    const int d = dst_out[(size_t)n * Kout + k] - n;
                    const float w2 = w * ASR_L2E;
                    if (d == 0 && wo0 == NI2) wo0 = w2;
                    else if (d == 1 && wo1 == NI2) wo1 = w2;
                    else if (d == 2 && wo2 == NI2) wo2 = w2;
                    else ok = false;
                }
            }
        }
        wb[s][0] = isB ? wo0 : wi0; wb[s][1] = isB ? wo1 : wi1; wb[s][2] = isB ? wo2 : wi2;
        term2[s] = valid ? fmaxf(term[n], p.neg_inf) * ASR_L2E : NI2;
        // batch padding / unreachable states (no in-arc, not the start state) never carry mass
        const bool alive = valid && (n == 0 || wi0 != NI2 || wi1 != NI2 || wi2 != NI2);
        dead[s] = !alive;
        labtab[n] = alive ? lab[s] : -1;
    }
    // slots whose valid states all share one label are reduced with DPP
    bool uni[S];
    int ul[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        ul[s] = __builtin_amdgcn_readfirstlane(lab[s]);
        const bool same = dead[s] || lab[s] == ul[s];
        uni[s] = __builtin_amdgcn_read_exec() == __ballot(same);      // wave-uniform
    }
    // the uniform slots must share ONE label (the blank); its posterior is one DPP sum per frame
    int ulab = -1;
#pragma unroll
    for (int s = S - 1; s >= 0; --s)
        if (uni[s]) ulab = ul[s];
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (uni[s] && ul[s] != ulab) ok = false;
    __syncthreads();
    // per-class gather lists over the remaining states: thread (class, half) scans half the states
    {
        const int c = lane, half = isB ? 1 : 0;
        const int nb = half ? (N + 1) / 2 : 0, ne = half ? N : (N + 1) / 2;
        int cnt = 0;
        for (int n = nb; n < ne; ++n) {
            const bool u = uni[0] ? (n % S) == 0 : false;
            bool slot_uni = u;
#pragma unroll
            for (int q = 1; q < S; ++q) slot_uni = slot_uni || (uni[q] && (n % S) == q);
            if (!slot_uni && labtab[n] == c) {
                if (cnt < BAND_LIST) lists[c][half][cnt] = (unsigned short)n;
                ++cnt;
            }
        }
        lcnt[c][half] = cnt;
    }
    __syncthreads();
    int mycnt = lcnt[lane][0] + lcnt[lane][1];
    if (lane >= C) mycnt = 0;
    if (mycnt > BAND_LIST) ok = false;
    if (!__syncthreads_and(ok)) {
        if (tid == 0) p.skip[b] = 0;
        return;
    }
    if (tid == 0) p.skip[b] = 1;
    int maxcnt = mycnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o, 64));
    maxcnt = __builtin_amdgcn_readfirstlane(maxcnt);
    for (int i = tid; i < 2 * (64 * S + 4); i += 128) (&gbuf[0][0])[i] = 0.f;

    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;
    float *grad_b = p.grad + (size_t)b * C;
    for (int t = len; t < p.T; ++t)                  // fst_utils.py:448
        for (int c = tid; c < C; c += 128) grad_b[(size_t)t * tstride + c] = 0.f;

    // The scan proper, instantiated per chain (ISB) and gather-list length (LN)
    // so that the loop bodies are single basic blocks with an exact VMEM count
    // per step (counted vmcnt waits, prefetch distance D steps).
    auto run = [&](auto isb_c, auto ln_c) {
        constexpr bool ISB = decltype(isb_c)::value;
        constexpr int LN = decltype(ln_c)::value;
        float *const G = gbuf[ISB ? 1 : 0];
        u32 lidx[LN];             // byte offsets into G; unused entries point at the zero slot
        {
            const int c0 = lcnt[lane][0];
#pragma unroll
            for (int i = 0; i < LN; ++i) {
                int n = 64 * S;
                if (i < mycnt) n = i < c0 ? lists[lane][0][i] : lists[lane][1][i - c0];
                lidx[i] = (u32)n * 4u;
            }
        }
        const int m = len >> 1, solo = len - 2 * m;
        // ---- buffers (bounds-checked: masked lanes / steps carry an out-of-range offset)
        const int Hw = (N + 63) / 64 * 64;           // ws row stride (asr_lattice_fwbw_workspace_bytes)
        const u32 ts4 = (u32)tstride * 4u, as4 = (u32)p.B * (u32)Hw * 4u;
        const u32 lp_bytes = (u32)(((size_t)p.T * p.B * C - (size_t)b * C) * 4);
        const u32 ws_bytes = (u32)(((size_t)(p.T + 2) * p.B * Hw - (size_t)b * Hw) * 4);
        const __amdgpu_buffer_rsrc_t lpR = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.lp) + (size_t)b * C, 0, lp_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t gradR =
            __builtin_amdgcn_make_buffer_rsrc(grad_b, 0, lp_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wsR = __builtin_amdgcn_make_buffer_rsrc(
            p.alphas + (size_t)b * Hw, 0, ws_bytes, 0x00020000);
        const u32 OOB = 0x80000000u;                 // every buffer is < 2^31 bytes (host check)
        const u32 c4 = lane < C ? (u32)lane * 4u : OOB;            // this lane's class column
        const u32 wn4 = lane * S < N ? (u32)lane * S * 4u : OOB;   // this lane's ws columns

        // ---- schedule of this chain: n0 steps before the meeting point, n1 after;
        // step j works on frame f(j) = j (alpha) or len-1-j (beta).  The first
        // phase is padded at the front, the second at the back, to whole rings.
        const int n0 = ISB ? m + solo : m, n1 = ISB ? m : m + solo;
        const int pad0 = (D - n0 % D) % D;
        auto frame_of = [&](int j) -> int { return ISB ? len - 1 - j : j; };
        auto row_off = [&](int j) -> u32 {            // lp row of step j
            const bool in = (j >= 0) & (j < n0 + n1);
            return in ? c4 + (u32)frame_of(j) * ts4 : OOB;
        };
        float R[D];                                   // ring of log-prob rows (raw)
#pragma unroll
        for (int u = 0; u < D; ++u)
            R[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lpR, row_off(u - pad0), 0, 0));
        auto gather = [&](float rowv, float (&em)[S]) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                em[s] = ASR_L2E * __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
                            lab[s] * 4, __builtin_bit_cast(int, rowv)));
        };

        float a[S];                                   // alpha_t / beta_{t+1}, log2 units
#pragma unroll
        for (int s = 0; s < S; ++s) a[s] = ISB ? term2[s] : ((lane == 0 && s == 0) ? 0.f : NI2);

        // one recurrence step with the emissions em of the step's frame
        auto advance = [&](const float (&em)[S], float (&out)[S]) {
            if constexpr (!ISB) {
                const float p3 = wave_shr1(a[S - 1], NI2), p2 = wave_shr1(a[S - 2], NI2);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float x0 = a[s] + wb[s][0];
                    const float x1 = (s >= 1 ? a[s >= 1 ? s - 1 : 0] : p3) + wb[s][1];
                    const float x2 = (s >= 2 ? a[s >= 2 ? s - 2 : 0] : (s == 1 ? p3 : p2)) + wb[s][2];
                    out[s] = em[s] + lse3_2(x0, x1, x2);
                }
            } else {
                float bt[S];
#pragma unroll
                for (int s = 0; s < S; ++s) bt[s] = a[s] + em[s];
                const float q0 = wave_shl1(bt[0], NI2), q1 = wave_shl1(bt[1], NI2);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float y0 = bt[s] + wb[s][0];
                    const float y1 = (s + 1 < S ? bt[s + 1 < S ? s + 1 : 0] : q0) + wb[s][1];
                    const float y2 = (s + 2 < S ? bt[s + 2 < S ? s + 2 : 0] : (s + 2 == S ? q0 : q1)) + wb[s][2];
                    out[s] = lse3_2(y0, y1, y2);
                }
            }
        };

        float em[S];
        gather(R[0], em);
        // ================= phase 0: up to the meeting point, keep the states in ws
        // alpha: slot t <- alpha_{t+1} after frame t;  beta: slot t <- beta_{t+1} before frame t
        for (int J0 = 0; J0 < n0 + pad0; J0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int j = J0 + u - pad0;
                const bool act = j >= 0;
                float nx[S];
                advance(em, nx);
                const u32x4 keep = {__builtin_bit_cast(u32, a[0]), __builtin_bit_cast(u32, a[1]),
                                    __builtin_bit_cast(u32, a[2]), __builtin_bit_cast(u32, a[3])};
#pragma unroll
                for (int s = 0; s < S; ++s) a[s] = act ? nx[s] : a[s];
                const u32x4 fresh = {__builtin_bit_cast(u32, a[0]), __builtin_bit_cast(u32, a[1]),
                                     __builtin_bit_cast(u32, a[2]), __builtin_bit_cast(u32, a[3])};
                const u32 slot = (u32)frame_of(j);
                __builtin_amdgcn_raw_buffer_store_b128(ISB ? keep : fresh, wsR,
                                                       act ? wn4 + slot * as4 : OOB, 0, 0);
                R[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                           lpR, row_off(j + D), 0, 0));
                __builtin_amdgcn_sched_barrier(0);     // keep the refills D steps ahead of their use
                gather(R[(u + 1) % D], em);
            }
        }
        // ================= meeting point: logZ = LSE_n(alpha_m + beta_m)
#pragma unroll
        for (int s = 0; s < S; ++s) xch[ISB ? 1 : 0][lane * S + s] = a[s];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ws rows of this chain are out
        __syncthreads();
        float logZ2;
        {
            float v[S], mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                v[s] = xch[0][lane * S + s] + xch[1][lane * S + s];
                if (lane * S + s >= N) v[s] = -INFINITY;
                mx = fmaxf(mx, v[s]);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) sum += __builtin_amdgcn_exp2f(v[s] - mx);
            sum = wave_sum(sum);
            logZ2 = mx + __builtin_amdgcn_logf(sum);
        }
        // ================= phase 1: finish the chain; posteriors against the other
        // chain's stored states (slot t holds alpha_{t+1} for t < m, beta_{t+1} for t >= m)
        auto ws_off = [&](int k) -> u32 {             // slot read by phase-1 step k
            return k < n1 ? wn4 + (u32)frame_of(n0 + k) * as4 : OOB;
        };
        u32x4 W[D];
#pragma unroll
        for (int u = 0; u < D; ++u) W[u] = __builtin_amdgcn_raw_buffer_load_b128(wsR, ws_off(u), 0, 0);
        float pend[LN];                               // gathered posteriors of the previous step
        float ptot = 0.f;
        u32 pgoff = OOB;
#pragma unroll
        for (int i = 0; i < LN; ++i) pend[i] = 0.f;
        auto flush = [&]() {                          // grad row of the previous step
            float r = lane == ulab ? ptot : 0.f;
#pragma unroll
            for (int i = 0; i < LN; ++i) r += pend[i];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, r), gradR, pgoff, 0, 0);
        };
        for (int K0 = 0; K0 < n1; K0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int k = K0 + u;
                const bool act = k < n1;
                float nx[S], gam[S];
                // (bit_cast of the whole vector: __builtin_bit_cast(float, W[u].y) on a vector ELEMENT
                // reads the vector's first dword with this hipcc)
                typedef __attribute__((ext_vector_type(4))) float f32x4v;
                const f32x4v wf = __builtin_bit_cast(f32x4v, W[u]);
                const float oth[S] = {wf.x, wf.y, wf.z, wf.w};
                advance(em, nx);
                // alpha wave: gamma from the NEW alpha_{t+1}; beta wave: from beta_{t+1} before the update
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float mine = ISB ? a[s] : nx[s];
                    gam[s] = __builtin_amdgcn_exp2f(mine + oth[s] - logZ2);
                    const bool live = (lane * S + s < N) & act;
                    gam[s] = live ? gam[s] : 0.f;
                    a[s] = act ? nx[s] : a[s];
                }
                flush();                               // previous step's row (its LDS reads are long back)
                // stage this step's posteriors: uniform slots -> one DPP sum, the rest -> gather lists
                // (scalar stores: a float4-typed store and the float-typed gather loads below would
                // be "no alias" for the compiler and the store gets dropped)
#pragma unroll
                for (int s = 0; s < S; ++s) G[lane * S + s] = gam[s];
#pragma unroll
                for (int i = 0; i < LN; ++i)
                    pend[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(G) + lidx[i]);
                {
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < S; ++q) v += uni[q] ? gam[q] : 0.f;
                    ptot = dpp_wave_sum(v);
                }
                pgoff = act ? c4 + (u32)frame_of(n0 + k) * ts4 : OOB;
                W[u] = __builtin_amdgcn_raw_buffer_load_b128(wsR, ws_off(k + D), 0, 0);
                R[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                           lpR, row_off(n0 + k + D), 0, 0));
                __builtin_amdgcn_sched_barrier(0);
                gather(R[(u + 1) % D], em);
            }
        }
        flush();
        // ================= totals
        {
            // alpha: logZ = LSE_n(alpha_len + terminal) (fst_utils.py:445); beta: from beta_0 (:476)
            float mx = -INFINITY, v[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float add = ISB ? ((lane == 0 && s == 0) ? 0.f : NI2) : term2[s];
                v[s] = lane * S + s < N ? a[s] + add : -INFINITY;
                mx = fmaxf(mx, v[s]);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) sum += __builtin_amdgcn_exp2f(v[s] - mx);
            sum = wave_sum(sum);
            float *dst = ISB ? p.logZ_bwd : p.logZ;
            if (lane == 0 && dst) dst[b] = (mx + __builtin_amdgcn_logf(sum)) * ASR_LN2;
        }
    };
    typedef std::integral_constant<bool, false> CA;
    typedef std::integral_constant<bool, true> CB;
    typedef std::integral_constant<int, BAND_LIST / 2> L8;
    typedef std::integral_constant<int, BAND_LIST> L16;
    if (isB) {
        if (maxcnt <= BAND_LIST / 2) run(CB(), L8()); else run(CB(), L16());
    } else {
        if (maxcnt <= BAND_LIST / 2) run(CA(), L8()); else run(CA(), L16());
    }
}



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
    // [T+2, B, round_up(N,64)] f32 (+1 slot: beta_len, +1: spare)
    const int64_t H = (N + 63) / 64 * 64;
    // + [B] band-kernel flags + [B] split-scatter flags
    return (int64_t)(T + 2) * B * H * (int64_t)sizeof(float) + (int64_t)B * 8 + 256;
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
    p.skip = nullptr;
    p.split = nullptr;
    bool split_scatter = false;

    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    size_t lds = (size_t)(2 * Npad + 2 * Cpad + 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int Kmax = Kin > Kout ? Kin : Kout;
    void (*kern)(FwbwParams);
    int nt;
    const size_t lds_mitm = (size_t)(4 * Npad + 4 * Cpad + 64) * sizeof(float);
    const bool fits32 = (size_t)T * B * C * 4 < (1ull << 31) &&
                        (size_t)(T + 2) * B * round_up(N, 64) * 4 < (1ull << 30);
    if (N <= 512 && Kmax <= 4 && lds_mitm <= 160 * 1024 && fits32) {
        // state-labelled fast path; a workgroup whose graph fails the entry
        // check runs the generic body inside the same launch
        const int H = round_up(N, 64);
        const size_t lds_sl = (size_t)(4 * H + 6 * Cpad + 64 + 2 * H) * sizeof(float);
        // Experimental (ASR_LATTICE_SPLIT_SCATTER=1): the per-class posterior sums leave
        // the sequential chain (FL == 2 + lattice_scatter_kernel).  Correct (same tests)
        // but slower today: 114 + 60 us vs 149 us fused on the B=512 mono numerator.
        const char *split_env = getenv("ASR_LATTICE_SPLIT_SCATTER");
        split_scatter = C <= H && T > 0 && split_env && split_env[0] == '1';
        if (split_scatter) {
            kern = Kmax <= 3 ? lattice_fwbw_sl_kernel<3, 8, 2> : lattice_fwbw_sl_kernel<4, 8, 2>;
            p.split = (int *)((char *)workspace + (size_t)(T + 2) * B * H * sizeof(float)) + B;
        } else if (C <= H)
            kern = Kmax <= 3 ? lattice_fwbw_sl_kernel<3, 8, 1> : lattice_fwbw_sl_kernel<4, 8, 1>;
        else
            kern = Kmax <= 3 ? lattice_fwbw_sl_kernel<3, 8, 0> : lattice_fwbw_sl_kernel<4, 8, 0>;
        nt = 2 * H;
        lds = lds_sl > lds_mitm ? lds_sl : lds_mitm;
        // Experimental (ASR_LATTICE_BAND=1): one-wave-per-chain kernel for band lattices.
        // Correct (same tests) but slower today: 180 us vs 151 us on the B=512 mono
        // numerator — a single wave per SIMD is issue-bound at ~1240 cycles per step.
        const char *band_env = getenv("ASR_LATTICE_BAND");
        if (band_env && band_env[0] == '1' && N <= 256 && C <= 64 && T > 0) {
            // band lattices (CTC chains) first; it flags the utterances it has done
            p.skip = (int *)((char *)workspace + (size_t)(T + 2) * B * H * sizeof(float));
            hipLaunchKernelGGL((lattice_fwbw_band_kernel<4, 8>), dim3(B), dim3(128), 0,
                               (hipStream_t)stream, p);
        }
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
    if (split_scatter) {
        constexpr int FPW = 4;
        const int nj = round_up(N, 64) / 64;
        void (*sk)(FwbwParams);
        switch (nj) {
            case 1: sk = lattice_scatter_kernel<1, FPW>; break;
            case 2: sk = lattice_scatter_kernel<2, FPW>; break;
            case 3: sk = lattice_scatter_kernel<3, FPW>; break;
            case 4: sk = lattice_scatter_kernel<4, FPW>; break;
            case 5: sk = lattice_scatter_kernel<5, FPW>; break;
            case 6: sk = lattice_scatter_kernel<6, FPW>; break;
            case 7: sk = lattice_scatter_kernel<7, FPW>; break;
            default: sk = lattice_scatter_kernel<8, FPW>; break;
        }
        hipLaunchKernelGGL(sk, dim3(B, (T + 4 * FPW - 1) / (4 * FPW)), dim3(256),
                           (size_t)4 * Cpad * sizeof(float), s, p);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
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
