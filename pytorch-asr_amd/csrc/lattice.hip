// Lattice scans for gfx950: log-semiring forward-backward (PathLogSumExp,
// reference att_speech/fst_utils.py:400-488) and the alpha-only scan with
// logsumexp / max reduction + best-path read-out (fst_utils.py:322-397,
// modules/decoders/advanced_decoder.py:546-554).
//
// Mapping: one workgroup per utterance; alpha/beta live in LDS (double
// buffered, one s_barrier per frame); the per-frame gradient row is
// accumulated in LDS (ds_add_f32) and streamed out coalesced, so every
// [t,b,:] row of the dense gradient is written exactly once.
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
    return (int64_t)T * B * N * (int64_t)sizeof(float) + 256;
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

    const int Npad = (N + 3) & ~3, Cpad = (C + 3) & ~3;
    const size_t lds = (size_t)(2 * Npad + 2 * Cpad + 64) * sizeof(float);
    if (lds > 160 * 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int Kmax = Kin > Kout ? Kin : Kout;
    void (*kern)(FwbwParams);
    int nt;
    if (N <= 1024 && Kmax <= 4) {
        kern = lattice_fwbw_kernel<4>;
        nt = round_up(N, 64);
    } else {
        kern = lattice_fwbw_kernel<0>;
        nt = N >= 1024 ? 1024 : round_up(N, 64);
    }
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return ASR_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, p);
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
