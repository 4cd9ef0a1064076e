// Forward-backward / Viterbi over GROUP-FACTORED graphs for gfx950.
//
// The CTC decoding graphs of the reference (build_ctc_mono_decoding_fst
// fst_utils.py:679-726, build_ctc_bigram_decoding_fst :729-835) — the
// denominator of the globally normalised CTC-G loss
// (advanced_decoder.py:473,497-500) and the search graph of FSTDecoder.decode
// (:542-554) — are dense but structured: every state s1 has a "next context"
// g(s1), every state s2 accepts an arc from EVERY s1 with g(s1) == h(s2), plus
// an optional extra self-loop, and all arcs into s2 carry label(s2), weight 0.
//   mono   : g = h = 0                      (all-to-all over S states)
//   bigram : s = (c,l): g = (l ? l : c), h = c; extra self-loop iff l != 0, c != l
// Hence
//   alpha'[s2] = lp[label s2] + LSE(R[h(s2)], self(s2) ? alpha[s2])
//   R[grp]     = LSE_{s1: g(s1) == grp} alpha[s1]
// and symmetrically for beta with Q[grp] = LSE_{s2: h(s2) == grp}(lp[label s2] + beta[s2]):
// ~S^2 + 2N transcendentals per frame instead of one per arc (122k for the
// 2401-state bigram graph).  Results equal the reference's sparse scan on the
// padded arc matrices up to fp32 summation order.
//
// One workgroup per utterance; alpha / beta, the group sums and the gradient row
// live in LDS; member lists are read from global memory (L2-resident, shared by
// the whole batch).
#include <cstdlib>
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

using namespace asr;

struct GroupedParams {
    const float *lp;            // [T,B,C]
    int T, B, C;
    const int32_t *lens;        // [B]
    int N, G, Wg, Wh;
    const int32_t *g_of;        // [N] group a state feeds
    const int32_t *h_of;        // [N] group a state accepts from
    const int32_t *label;       // [N]
    const int32_t *selfx;       // [N] 1: extra self-loop
    const int32_t *uniq;        // [N] 1: no other state carries this label
    const int32_t *mem_g;       // [G,Wg] states with g == grp (ascending, -1 padded)
    const int32_t *mem_h;       // [G,Wh] states with h == grp (ascending, -1 padded)
    const float *term;          // [N] terminal log-weights
    float neg_inf;
    float *logZ, *grad, *logZ_bwd;
    float *alphas;              // [T+1,B,N] alpha after t frames
    float *score;               // viterbi
    int32_t *best_il;           // [T,B]
    int32_t *rarg;              // [T,B,G] arg-max member of every group (viterbi)
    uint8_t *bpself;            // [T,B,N] 1: best predecessor is the state itself
    int accumulate;             // fwbw: grad += occupancies (rows past an utterance's end untouched)
};

constexpr int LPG = 16;         // lanes cooperating on one group sum

__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}

// LSE over the members of every group: val(s) for s in mem[grp,:].
// Threads [0, G*LPG) take part; result in out[grp].
template <typename F>
__device__ __forceinline__ void group_lse(const int32_t *mem, int G, int W, float neg_inf,
                                          float *out, F val) {
    const int tid = threadIdx.x;
    if (tid < G * LPG) {
        const int grp = tid / LPG, l = tid % LPG;
        const int32_t *mm = mem + (size_t)grp * W;
        float m = neg_inf;
        for (int j = l; j < W; j += LPG) {
            const int s = mm[j];
            if (s >= 0) m = fmaxf(m, val(s));
        }
#pragma unroll
        for (int o = LPG / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, LPG));
        float sum = 0.f;
        for (int j = l; j < W; j += LPG) {
            const int s = mm[j];
            if (s >= 0) sum += __expf(val(s) - m);
        }
#pragma unroll
        for (int o = LPG / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, LPG);
        if (l == 0) out[grp] = sum > 0.f ? m + __logf(sum) : neg_inf;
    }
}

__global__ __launch_bounds__(1024) void grouped_fwbw_kernel(GroupedParams p) {
    extern __shared__ float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int N = p.N, G = p.G, C = p.C;
    float *a0 = smem;                    // [N] alpha / beta (current)
    float *a1 = a0 + N;                  // [N] next / u = lp + beta
    float *R = a1 + N;                   // [G]
    float *row = R + ((G + 3) & ~3);     // [C] gradient row
    float *red = row + ((C + 3) & ~3);   // [32]
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;
    const float *lp_b = p.lp + (size_t)b * C;
    float *grad_b = p.grad + (size_t)b * C;
    const size_t astride = (size_t)p.B * N;
    float *al_b = p.alphas + (size_t)b * N;

    if (!p.accumulate)
        for (int t = len; t < p.T; ++t)                   // fst_utils.py:448
            for (int c = tid; c < C; c += NT) grad_b[(size_t)t * tstride + c] = 0.f;
    for (int n = tid; n < N; n += NT) {
        const float v = n == 0 ? 0.f : p.neg_inf;         // start state 0 (fst_utils.py:246)
        a0[n] = v;
        al_b[n] = v;
    }
    for (int c = tid; c < C; c += NT) row[c] = 0.f;
    __syncthreads();

    // ---------------- forward ----------------
    float *cur = a0, *nxt = a1;
    for (int t = 0; t < len; ++t) {
        const float *lrow = lp_b + (size_t)t * tstride;
        group_lse(p.mem_g, G, p.Wg, p.neg_inf, R, [&](int s) { return cur[s]; });
        __syncthreads();
        for (int n = tid; n < N; n += NT) {
            const float r = R[p.h_of[n]];
            const float v = lrow[p.label[n]] + (p.selfx[n] ? lse2(r, cur[n]) : r);
            nxt[n] = v;
            al_b[(size_t)(t + 1) * astride + n] = v;
        }
        __syncthreads();
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    float logZ;
    {
        float m = -INFINITY;
        for (int n = tid; n < N; n += NT) m = fmaxf(m, cur[n] + p.term[n]);
        m = block_max(m, red);
        float s = 0.f;
        for (int n = tid; n < N; n += NT) s += __expf(cur[n] + p.term[n] - m);
        s = block_sum(s, red);
        logZ = m + __logf(s);
        if (tid == 0) p.logZ[b] = logZ;
    }
    __syncthreads();

    // ---------------- backward ----------------
    // cur = beta_{t+1}; nxt = u = lp_t[label] + beta_{t+1}
    for (int n = tid; n < N; n += NT) cur[n] = p.term[n];
    __syncthreads();
    for (int t = len - 1; t >= 0; --t) {
        const float *lrow = lp_b + (size_t)t * tstride;
        const float *arow = al_b + (size_t)(t + 1) * astride;
        for (int n = tid; n < N; n += NT) {
            const float bt = cur[n];
            const int lab = p.label[n];
            nxt[n] = lrow[lab] + bt;
            // posterior of being in state n after frame t
            const float o = __expf(arow[n] + bt - logZ);
            if (o != 0.f) {
                if (p.uniq[n]) row[lab] = o;
                else atomicAdd(&row[lab], o);
            }
        }
        __syncthreads();
        group_lse(p.mem_h, G, p.Wh, p.neg_inf, R, [&](int s) { return nxt[s]; });
        {   // stream the finished gradient row out while the group sums settle
            float *gout = grad_b + (size_t)t * tstride;
            if (p.accumulate)
                for (int c = tid; c < C; c += NT) gout[c] += row[c];
            else
                for (int c = tid; c < C; c += NT) gout[c] = row[c];
        }
        __syncthreads();
        for (int c = tid; c < C; c += NT) row[c] = 0.f;
        for (int n = tid; n < N; n += NT) {
            const float q = R[p.g_of[n]];
            cur[n] = p.selfx[n] ? lse2(q, nxt[n]) : q;
        }
        __syncthreads();
    }
    if (p.logZ_bwd && tid == 0) p.logZ_bwd[b] = cur[0];   // logsumexp(alpha_0 + beta_0) (:476)
}

// The same scan with everything a frame needs already on the CU (round 3; the kernel above is
// kept for graphs whose groups are wider than 64 states).  The first version spent ~10 k cycles
// per frame and direction on one CU per utterance, almost all of it waiting: the member lists,
// h / g / label / self-loop flags of every state and the frame's log-probs were fetched from
// global memory inside the dependent chain, frame after frame.  Here a thread keeps the
// description of its (<= SPT) states in registers, a lane of a group's 16-lane team keeps its
// (<= 4) members of both lists in registers, and the log-probs / stored alphas of the NEXT
// frame are loaded while the current one is computed.
// LG lanes share a group sum (64 / LG members per lane): 16 for 1024-thread workgroups, 8 for
// the 512-thread ones that large batches get (four of them fit a CU, two of the others)
template <int SPT, int LG = LPG>
__global__ __launch_bounds__(1024) void grouped_fwbw_fast_kernel(GroupedParams p) {
    constexpr int MPL = 64 / LG;
    extern __shared__ float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int N = p.N, G = p.G, C = p.C;
    float *a0 = smem;                    // [N] alpha / beta (current)
    float *a1 = a0 + N;                  // [N] next / u = lp + beta
    float *R = a1 + N;                   // [G]
    float *row = R + ((G + 3) & ~3);     // [C] gradient row
    float *red = row + ((C + 3) & ~3);   // [32]
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;
    const float *lp_b = p.lp + (size_t)b * C;
    float *grad_b = p.grad + (size_t)b * C;
    const size_t astride = (size_t)p.B * N;
    float *al_b = p.alphas + (size_t)b * N;

    // this thread's states
    int sh[SPT], sg[SPT], slab[SPT];
    bool sv[SPT], sself[SPT], suniq[SPT];
    float sterm[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int n = tid + k * NT;
        sv[k] = n < N;
        const int nn = sv[k] ? n : 0;
        sh[k] = p.h_of[nn]; sg[k] = p.g_of[nn]; slab[k] = p.label[nn];
        sself[k] = p.selfx[nn] != 0; suniq[k] = p.uniq[nn] != 0;
        sterm[k] = p.term[nn];
    }
    // this lane's members of its group (both lists)
    const bool glane = tid < G * LG;
    const int grp = tid / LG, gl = tid % LG;
    int mg[MPL], mh[MPL];
#pragma unroll
    for (int j = 0; j < MPL; ++j) {
        const int idx = gl + LG * j;
        mg[j] = (glane && idx < p.Wg) ? p.mem_g[(size_t)grp * p.Wg + idx] : -1;
        mh[j] = (glane && idx < p.Wh) ? p.mem_h[(size_t)grp * p.Wh + idx] : -1;
    }
    auto group_lse_reg = [&](const float *val, const int *mem) -> float {
        float v[MPL], m = p.neg_inf;
#pragma unroll
        for (int j = 0; j < MPL; ++j) {
            v[j] = mem[j] >= 0 ? val[mem[j]] : p.neg_inf;
            m = fmaxf(m, v[j]);
        }
#pragma unroll
        for (int o = LG / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, LG));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MPL; ++j) sum += mem[j] >= 0 ? __expf(v[j] - m) : 0.f;
#pragma unroll
        for (int o = LG / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, LG);
        return sum > 0.f ? m + __logf(sum) : p.neg_inf;
    };
    auto lpload = [&](int t, float *o) {
        const float *lrow = lp_b + (size_t)t * tstride;
#pragma unroll
        for (int k = 0; k < SPT; ++k) o[k] = sv[k] ? lrow[slab[k]] : 0.f;
    };

    if (!p.accumulate)
        for (int t = len; t < p.T; ++t)                   // fst_utils.py:448
            for (int c = tid; c < C; c += NT) grad_b[(size_t)t * tstride + c] = 0.f;
#pragma unroll
    for (int k = 0; k < SPT; ++k)
        if (sv[k]) {
            const int n = tid + k * NT;
            const float v = n == 0 ? 0.f : p.neg_inf;     // start state 0 (fst_utils.py:246)
            a0[n] = v;
            al_b[n] = v;
        }
    for (int c = tid; c < C; c += NT) row[c] = 0.f;
    float lpc[SPT], lpn[SPT];
    if (len > 0) lpload(0, lpc);
    __syncthreads();

    // ---------------- forward ----------------
    float *cur = a0, *nxt = a1;
    for (int t = 0; t < len; ++t) {
        if (t + 1 < len) lpload(t + 1, lpn);
        if (glane) {
            const float r = group_lse_reg(cur, mg);
            if (gl == 0) R[grp] = r;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (sv[k]) {
                const int n = tid + k * NT;
                const float r = R[sh[k]];
                const float v = lpc[k] + (sself[k] ? lse2(r, cur[n]) : r);
                nxt[n] = v;
                al_b[(size_t)(t + 1) * astride + n] = v;
            }
        __syncthreads();
        float *tmp = cur; cur = nxt; nxt = tmp;
#pragma unroll
        for (int k = 0; k < SPT; ++k) lpc[k] = lpn[k];
    }
    float logZ;
    {
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (sv[k]) m = fmaxf(m, cur[tid + k * NT] + sterm[k]);
        m = block_max(m, red);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (sv[k]) s += __expf(cur[tid + k * NT] + sterm[k] - m);
        s = block_sum(s, red);
        logZ = m + __logf(s);
        if (tid == 0) p.logZ[b] = logZ;
    }
    __syncthreads();

    // ---------------- backward ----------------
    // cur = beta_{t+1}; nxt = u = lp_t[label] + beta_{t+1}
#pragma unroll
    for (int k = 0; k < SPT; ++k)
        if (sv[k]) cur[tid + k * NT] = sterm[k];
    float ac[SPT], an[SPT];
    auto aload = [&](int t, float *o) {                    // alpha after t frames
        const float *arow = al_b + (size_t)t * astride;
#pragma unroll
        for (int k = 0; k < SPT; ++k) o[k] = sv[k] ? arow[tid + k * NT] : 0.f;
    };
    if (len > 0) {
        lpload(len - 1, lpc);
        aload(len, ac);
    }
    __syncthreads();
    for (int t = len - 1; t >= 0; --t) {
        if (t > 0) {
            lpload(t - 1, lpn);
            aload(t, an);
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (sv[k]) {
                const int n = tid + k * NT;
                const float bt = cur[n];
                nxt[n] = lpc[k] + bt;
                // posterior of being in state n after frame t
                const float o = __expf(ac[k] + bt - logZ);
                if (o != 0.f) {
                    if (suniq[k]) row[slab[k]] = o;
                    else atomicAdd(&row[slab[k]], o);
                }
            }
        __syncthreads();
        if (glane) {
            const float r = group_lse_reg(nxt, mh);
            if (gl == 0) R[grp] = r;
        }
        {   // the finished gradient row leaves (and is cleared for the next frame)
            float *gout = grad_b + (size_t)t * tstride;
            if (p.accumulate) {
                for (int c = tid; c < C; c += NT) {
                    gout[c] += row[c];
                    row[c] = 0.f;
                }
            } else {
                for (int c = tid; c < C; c += NT) {
                    gout[c] = row[c];
                    row[c] = 0.f;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (sv[k]) {
                const int n = tid + k * NT;
                const float q = R[sg[k]];
                cur[n] = sself[k] ? lse2(q, nxt[n]) : q;
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPT; ++k) { lpc[k] = lpn[k]; ac[k] = an[k]; }
    }
    if (p.logZ_bwd && tid == 0) p.logZ_bwd[b] = cur[0];   // logsumexp(alpha_0 + beta_0) (:476)
}

template <bool VITERBI>
__global__ __launch_bounds__(1024) void grouped_forward_kernel(GroupedParams p) {
    extern __shared__ float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int N = p.N, G = p.G, C = p.C;
    float *a0 = smem, *a1 = a0 + N;
    float *R = a1 + N;
    int *Rarg = (int *)(R + ((G + 3) & ~3));
    float *red = (float *)(Rarg + ((G + 3) & ~3));
    int *redi = (int *)(red + 32);
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.T ? p.T : len);
    const size_t tstride = (size_t)p.B * C;
    const float *lp_b = p.lp + (size_t)b * C;
    const bool want_path = VITERBI && p.best_il != nullptr;

    for (int n = tid; n < N; n += NT) a0[n] = n == 0 ? 0.f : p.neg_inf;
    if (want_path)
        for (int t = len + tid; t < p.T; t += NT) p.best_il[(size_t)t * p.B + b] = 0;
    __syncthreads();

    float *cur = a0, *nxt = a1;
    for (int t = 0; t < len; ++t) {
        const float *lrow = lp_b + (size_t)t * tstride;
        if (VITERBI) {
            // best member of every group; ties -> lowest state id (the reference's
            // torch.max takes the first of the arcs sorted by source state)
            if (tid < G * LPG) {
                const int grp = tid / LPG, l = tid % LPG;
                const int32_t *mm = p.mem_g + (size_t)grp * p.Wg;
                float m = -INFINITY;
                int arg = 0x7fffffff;
                for (int j = l; j < p.Wg; j += LPG) {
                    const int s = mm[j];
                    if (s >= 0) {
                        const float v = cur[s];
                        if (v > m || (v == m && s < arg)) { m = v; arg = s; }
                    }
                }
#pragma unroll
                for (int o = LPG / 2; o > 0; o >>= 1) {
                    const float om = __shfl_xor(m, o, LPG);
                    const int oa = __shfl_xor(arg, o, LPG);
                    if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
                }
                if (l == 0) {
                    R[grp] = m;
                    Rarg[grp] = arg;
                    if (want_path) p.rarg[((size_t)t * p.B + b) * G + grp] = arg;
                }
            }
        } else {
            group_lse(p.mem_g, G, p.Wg, p.neg_inf, R, [&](int s) { return cur[s]; });
        }
        __syncthreads();
        for (int n = tid; n < N; n += NT) {
            const int h = p.h_of[n];
            const float r = R[h];
            float v;
            if (VITERBI) {
                // predecessor: best member of the group, or the state itself
                // through its extra self-loop (ties -> lower source state id)
                bool self = false;
                v = r;
                if (p.selfx[n]) {
                    const float x = cur[n];
                    if (x > r || (x == r && n < Rarg[h])) { v = x; self = true; }
                }
                if (want_path) p.bpself[((size_t)t * p.B + b) * N + n] = self ? 1 : 0;
            } else {
                v = p.selfx[n] ? lse2(r, cur[n]) : r;
            }
            nxt[n] = lrow[p.label[n]] + v;
        }
        __syncthreads();
        float *tmp = cur; cur = nxt; nxt = tmp;
    }

    if (VITERBI) {
        float best = -INFINITY;
        int arg = 0x7fffffff;
        for (int n = tid; n < N; n += NT) {
            const float v = cur[n] + p.term[n];
            if (v > best) { best = v; arg = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
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
                int st = arg;                          // state after frame len-1
                for (int t = len - 1; t >= 0; --t) {
                    p.best_il[(size_t)t * p.B + b] = p.label[st];
                    const size_t tb = (size_t)t * p.B + b;
                    if (!p.bpself[tb * N + st]) st = p.rarg[tb * G + p.h_of[st]];
                }
            }
        }
    } else {
        float m = -INFINITY;
        for (int n = tid; n < N; n += NT) m = fmaxf(m, cur[n] + p.term[n]);
        m = block_max(m, red);
        float s = 0.f;
        for (int n = tid; n < N; n += NT) s += __expf(cur[n] + p.term[n] - m);
        s = block_sum(s, red);
        if (tid == 0) p.score[b] = m + __logf(s);
    }
}

inline int round_up64(int v) { return (v + 63) / 64 * 64; }

inline bool bad_common(const float *lp, int T, int B, int C, const int32_t *lens, int N, int G,
                       int Wg, int Wh, const void *a, const void *b2, const void *c,
                       const void *d, const void *e, const void *f, const void *g2,
                       const void *h) {
    if (T < 0 || B < 0 || C <= 0 || N <= 0 || G <= 0 || Wg <= 0 || Wh <= 0) return true;
    if (G * LPG > 1024) return true;
    if (B > 0 && (!lens || !a || !b2 || !c || !d || !e || !f || !g2 || !h)) return true;
    if (B > 0 && T > 0 && !lp) return true;
    return false;
}

}  // namespace

extern "C" int64_t asr_lattice_grouped_workspace_bytes(int T, int B, int N, int G) {
    if (T < 0 || B < 0 || N < 0 || G < 0) return -1;
    // alphas f32 [T+1,B,N]  |  viterbi: rarg i32 [T,B,G] + bpself u8 [T,B,N]
    const int64_t fw = (int64_t)(T + 1) * B * N * 4;
    const int64_t vt = (int64_t)T * B * G * 4 + (int64_t)T * B * N + 64;
    return (fw > vt ? fw : vt) + 256;
}

extern "C" int asr_lattice_grouped_fwbw_acc_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, int accumulate, float *out_logZ, float *out_grad, float *out_logZ_bwd,
    void *workspace, int64_t workspace_bytes, void *stream) {
    if (bad_common(lp, T, B, C, lens, N, G, Wg, Wh, g_of, h_of, label, selfx, uniq, mem_g,
                   mem_h, term))
        return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!out_logZ || (T > 0 && !out_grad) || !(neg_inf < 0.f)) return ASR_EINVAL;
    if (!workspace || workspace_bytes < asr_lattice_grouped_workspace_bytes(T, B, N, G))
        return ASR_EINVAL;
    GroupedParams p = {};
    p.lp = lp; p.T = T; p.B = B; p.C = C; p.lens = lens;
    p.N = N; p.G = G; p.Wg = Wg; p.Wh = Wh;
    p.g_of = g_of; p.h_of = h_of; p.label = label; p.selfx = selfx; p.uniq = uniq;
    p.mem_g = mem_g; p.mem_h = mem_h; p.term = term; p.neg_inf = neg_inf;
    p.logZ = out_logZ; p.grad = out_grad; p.logZ_bwd = out_logZ_bwd;
    p.alphas = (float *)workspace;
    p.accumulate = accumulate ? 1 : 0;
    const size_t lds = (size_t)(2 * N + ((G + 3) & ~3) + ((C + 3) & ~3) + 64) * sizeof(float);
    if (lds > 160 * 1024) return ASR_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)grouped_fwbw_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ASR_EUNSUPPORTED;
    int nt = N >= 1024 ? 1024 : round_up64(N);
    if (nt < G * LPG) nt = round_up64(G * LPG);
    const int spt = (N + nt - 1) / nt;
    static const bool fast_on = !(getenv("ASR_GROUPED_FAST") && getenv("ASR_GROUPED_FAST")[0] == '0');
    // 512-thread workgroups (five states per thread, eight lanes per group sum): four fit a CU
    // instead of two, so 768 utterances are resident at once instead of in one and a half
    // rounds (3.87 -> 3.06 ms), and at 256 utterances they are no slower (1.36 -> 1.32 ms).
    // ASR_GROUPED_NT=1024: the 1024-thread arrangement (A/B switch)
    static const char *nt_env = getenv("ASR_GROUPED_NT");
    const bool small_wg = !(nt_env && atoi(nt_env) == 1024);
    if (fast_on && small_wg && N > 512 && N <= 5 * 512 && Wg <= 64 && Wh <= 64 && G * 8 <= 512 &&
        lds <= 64 * 1024) {
        const int sp = (N + 511) / 512;
        void (*kern)(GroupedParams) = sp <= 2 ? grouped_fwbw_fast_kernel<2, 8>
                                      : sp == 3 ? grouped_fwbw_fast_kernel<3, 8>
                                      : sp == 4 ? grouped_fwbw_fast_kernel<4, 8> : grouped_fwbw_fast_kernel<5, 8>;
        hipLaunchKernelGGL(kern, dim3(B), dim3(512), lds, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    if (fast_on && Wg <= 4 * LPG && Wh <= 4 * LPG && spt <= 4 && nt <= 1024 && lds <= 64 * 1024) {
        void (*kern)(GroupedParams) = spt <= 1 ? grouped_fwbw_fast_kernel<1>
                                      : spt == 2 ? grouped_fwbw_fast_kernel<2>
                                      : spt == 3 ? grouped_fwbw_fast_kernel<3> : grouped_fwbw_fast_kernel<4>;
        hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    hipLaunchKernelGGL(grouped_fwbw_kernel, dim3(B), dim3(nt), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_lattice_grouped_fwbw_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, float *out_logZ, float *out_grad, float *out_logZ_bwd, void *workspace,
    int64_t workspace_bytes, void *stream) {
    return asr_lattice_grouped_fwbw_acc_f32(lp, T, B, C, lens, N, G, Wg, Wh, g_of, h_of, label, selfx, uniq,
                                            mem_g, mem_h, term, neg_inf, 0, out_logZ, out_grad, out_logZ_bwd,
                                            workspace, workspace_bytes, stream);
}

extern "C" int asr_lattice_grouped_forward_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, int viterbi, float *out_score, int32_t *out_best_il, void *workspace,
    int64_t workspace_bytes, void *stream) {
    if (bad_common(lp, T, B, C, lens, N, G, Wg, Wh, g_of, h_of, label, selfx, uniq, mem_g,
                   mem_h, term))
        return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!out_score || !(neg_inf < 0.f)) return ASR_EINVAL;
    const bool want_path = viterbi && out_best_il;
    if (want_path && T > 0 &&
        (!workspace || workspace_bytes < asr_lattice_grouped_workspace_bytes(T, B, N, G)))
        return ASR_EINVAL;
    GroupedParams p = {};
    p.lp = lp; p.T = T; p.B = B; p.C = C; p.lens = lens;
    p.N = N; p.G = G; p.Wg = Wg; p.Wh = Wh;
    p.g_of = g_of; p.h_of = h_of; p.label = label; p.selfx = selfx; p.uniq = uniq;
    p.mem_g = mem_g; p.mem_h = mem_h; p.term = term; p.neg_inf = neg_inf;
    p.score = out_score;
    p.best_il = want_path ? out_best_il : nullptr;
    p.rarg = (int32_t *)workspace;
    p.bpself = (uint8_t *)workspace + (size_t)T * B * G * 4;
    const size_t lds = (size_t)(2 * N + 2 * ((G + 3) & ~3) + 128) * sizeof(float);
    if (lds > 160 * 1024) return ASR_EUNSUPPORTED;
    void (*kern)(GroupedParams) =
        viterbi ? grouped_forward_kernel<true> : grouped_forward_kernel<false>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return ASR_EUNSUPPORTED;
    int nt = N >= 1024 ? 1024 : round_up64(N);
    if (nt < G * LPG) nt = round_up64(G * LPG);
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
