// Decode-step kernels of the TCN / local-attention decoder:
//
//  asr_tcn_attention_step_f32 — LocalAttention.forward + the context reduction of
//      AttentionDecoderTCN.enc_step (reference att_speech/modules/tcn.py:193-230,
//      :465-474) for every live hypothesis in ONE launch: the per-hypothesis
//      location filter slid over the parent's previous alignment, + projected encoder
//      frame + global LM term, tanh, score, temperature, padding mask, softmax over the
//      encoder frames, context = sum_t a_t enc_t.  The reference runs this as a grouped
//      conv1d with B*beam groups and ~12 elementwise / reduction launches.
//
//  asr_beam_step_f32 — BeamSearch.step (reference att_speech/modules/beam_search.py:
//      58-124, 147-175) for every utterance in ONE launch and WITHOUT a host read-back:
//      log-softmax, running scores, best-EOS bookkeeping with length normalisation
//      (incl. the reference's indexing quirk), top-`beam` over beam*(C-1) extensions,
//      hypothesis re-indexing, and the all-finished flag; once the flag is set further
//      calls are no-ops, so a host that polls it every few steps gets the results of
//      the reference's step-exact stop.
//
// Plain extern "C" (include/asr_amd.h); fp32 throughout.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

using namespace asr;

constexpr int ATT_NT = 256;
constexpr int KF = 32;        // taps of the location filter (LocalAttention kernel_size)

struct AttParams {
    const float *eproj, *enc, *filt, *glob, *w_score, *att_prev;
    const int32_t *enc_lens, *parent;
    int T, B, beam, A, E;
    float b_score, temperature;
    float *att_new, *context;
};

// one workgroup per hypothesis
__global__ __launch_bounds__(ATT_NT) void tcn_attention_step_kernel(AttParams p) {
    extern __shared__ float smem[];
    const int h = blockIdx.x, tid = threadIdx.x;
    const int u = h / p.beam;
    const int T = p.T, A = p.A, E = p.E;
    float *aprev = smem;                  // [KF - 1 zeros][T]
    float *anew = smem + (KF - 1) + T;    // [T]
    float *red = anew + T;                // [32]
    const int src = p.parent ? p.parent[h] : h;
    const float *ap = p.att_prev + (size_t)src * T;
    for (int i = tid; i < KF - 1 + T; i += ATT_NT) aprev[i] = i < KF - 1 ? 0.f : ap[i - (KF - 1)];
    __syncthreads();

    const float *__restrict__ filt = p.filt + (size_t)h * A * KF;      // wave-uniform addresses
    const float *__restrict__ glob = p.glob + (size_t)h * A;
    const int len = p.enc_lens[u];
    float emax = -INFINITY;
    for (int t0 = 0; t0 < T; t0 += ATT_NT) {
        const int t = t0 + tid;
        float win[KF];
        const bool on = t < T;
#pragma unroll
        for (int j = 0; j < KF; ++j) win[j] = on ? aprev[t + j] : 0.f;   // a_prev[t - (KF-1) + j]
        const float *ep = p.eproj + ((size_t)(on ? t : 0) * p.B + u) * A;
        float e = 0.f;
        for (int c = 0; c < A; ++c) {
            float hid = ep[c] + glob[c];
            const float *f = filt + c * KF;
#pragma unroll
            for (int j = 0; j < KF; ++j) hid = fmaf(win[j], f[j], hid);
            // tanh(x) = 1 - 2 / (exp(2x) + 1), saturating cleanly at +-1
            const float ex = __expf(2.f * hid);
            const float th = 1.f - 2.f / (ex + 1.f);
            e = fmaf(p.w_score[c], th, e);
        }
        e = (e + p.b_score) * p.temperature + (t >= len ? -1e5f : 0.f);
        if (on) {
            anew[t] = e;
            emax = fmaxf(emax, e);
        }
    }
    emax = block_max(emax, red);
    float sum = 0.f;
    for (int t = tid; t < T; t += ATT_NT) {
        const float v = __expf(anew[t] - emax);
        anew[t] = v;
        sum += v;
    }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    float *out = p.att_new + (size_t)h * T;
    for (int t = tid; t < T; t += ATT_NT) {
        const float v = anew[t] * inv;
        anew[t] = v;
        out[t] = v;
    }
    __syncthreads();
    // context[e] = sum_t a[t] enc[t, u, e]   (coalesced over e; the beam's hypotheses of one
    // utterance re-read the same rows from L2)
    for (int e0 = tid; e0 < E; e0 += ATT_NT) {
        const float *col = p.enc + (size_t)u * E + e0;
        const size_t ts = (size_t)p.B * E;
        float acc0 = 0.f, acc1 = 0.f;
        int t = 0;
        for (; t + 1 < T; t += 2) {
            acc0 = fmaf(anew[t], col[(size_t)t * ts], acc0);
            acc1 = fmaf(anew[t + 1], col[(size_t)(t + 1) * ts], acc1);
        }
        if (t < T) acc0 = fmaf(anew[t], col[(size_t)t * ts], acc0);
        p.context[(size_t)h * E + e0] = acc0 + acc1;
    }
}

// ---------------------------------------------------------------------------
struct BeamParams {
    const float *logits, *scores_in;
    float *scores_out;
    const int32_t *est_in;
    int32_t *est_out;
    int step, B, beam, C, Lcap;
    float len_div;
    int32_t *finished_count, *best_len, *best_tokens;
    float *best_score;
    int32_t *new_input, *parent, *done, *unfinished;
};

constexpr int BEAM_NT = 128;
constexpr int BEAM_MAX = 32;         // hypotheses per utterance
constexpr int CAND_PER_THREAD = 16;  // BEAM_NT * 16 >= beam * (C - 1)

// log-softmax of one row of C logits + the running score, by one wave-sized group of lanes
__device__ __forceinline__ float row_logZ(const float *row, int C, int lane) {
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - m);
    s = wave_sum(s);
    return m + __logf(s);
}

// one workgroup per utterance
__global__ __launch_bounds__(BEAM_NT) void beam_step_kernel(BeamParams p) {
    __shared__ float lz[BEAM_MAX];           // log-partition of each hypothesis' row
    __shared__ float red_v[BEAM_NT / 64 * 2];
    __shared__ int red_i[BEAM_NT / 64 * 2];
    __shared__ float sel_v[BEAM_MAX];
    __shared__ int sel_i[BEAM_MAX];
    if (*p.done) return;                     // every utterance finished in an earlier step
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beam = p.beam, C = p.C, Cm = C - 1;
    const int h0 = b * beam;
    for (int k = wave; k < beam; k += BEAM_NT / 64) {
        const float z = row_logZ(p.logits + (size_t)(h0 + k) * C, C, lane);
        if (lane == 0) lz[k] = z;
    }
    __syncthreads();
    auto gscore = [&](int k, int c) -> float {       // global score of extension (beam k, class c)
        return (p.logits[(size_t)(h0 + k) * C + c] - lz[k]) + p.scores_in[h0 + k];
    };

    // ---- best finished hypothesis (beam_search.py:58-81) --------------------------
    if (p.step > 0 && wave == 0) {
        // quirk kept: `is_eos_best` is evaluated for the hypothesis whose FLAT index is the
        // utterance id b (reference :73), not for the utterance's own beams
        const float *qrow = p.logits + (size_t)b * C;
        float mo = -INFINITY;
        for (int c = lane; c < Cm; c += 64) mo = fmaxf(mo, qrow[c]);
        mo = wave_max(mo);
        const bool eos_best = qrow[Cm] > mo;           // argmax == C-1: first maximum wins
        // best length-normalised EOS score among the utterance's beams (first maximum)
        const float ln = p.len_div;      // step ** length_normalization, computed by the host
        float bestn = -INFINITY, bestr = 0.f;
        int bi = 0;
        for (int k = 0; k < beam; ++k) {
            const float raw = gscore(k, Cm), nrm = raw / ln;
            if (nrm > bestn) { bestn = nrm; bestr = raw; bi = k; }
        }
        if (eos_best && p.finished_count[b] <= beam) {
            if (lane == 0) p.finished_count[b] += 1;
            if (p.best_score[b] < bestn) {
                // the reference's aliased lists keep the RAW score (:76-78)
                const int32_t *src = p.est_in + (size_t)(h0 + bi) * p.Lcap;
                for (int i = lane; i < p.step; i += 64) p.best_tokens[(size_t)b * p.Lcap + i] = src[i];
                if (lane == 0) { p.best_score[b] = bestr; p.best_len[b] = p.step; }
            }
        }
    }
    __syncthreads();

    // ---- top-`beam` of the non-EOS extensions (:83-98, 126-133) ---------------------
    const int ncand = (p.step == 0 ? 1 : beam) * Cm;        // first step: beam 0 only
    float cv[CAND_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CAND_PER_THREAD; ++i) {
        const int idx = tid + i * BEAM_NT;
        cv[i] = idx < ncand ? gscore(idx / Cm, idx % Cm) : -INFINITY;
    }
    for (int r = 0; r < beam; ++r) {
        float bv = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < CAND_PER_THREAD; ++i) {
            const int idx = tid + i * BEAM_NT;
            if (idx < ncand && (cv[i] > bv || (cv[i] == bv && idx < bidx)) && cv[i] > -INFINITY) {
                bv = cv[i];
                bidx = idx;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ov > bv || (ov == bv && oi < bidx)) { bv = ov; bidx = oi; }
        }
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bidx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < BEAM_NT / 64; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bidx)) { bv = red_v[w]; bidx = red_i[w]; }
            sel_v[r] = bv;
            sel_i[r] = bidx;
        }
        __syncthreads();
        const int win = sel_i[r];
#pragma unroll
        for (int i = 0; i < CAND_PER_THREAD; ++i)
            if (tid + i * BEAM_NT == win) cv[i] = -INFINITY;      // taken
    }
    // fewer candidates than beams (:86-97): -inf scores, the last index repeated
    if (tid == 0) {
        int last = 0;
        for (int r = 0; r < beam; ++r) {
            if (sel_i[r] == 0x7fffffff) { sel_i[r] = last; sel_v[r] = -INFINITY; }
            else last = sel_i[r];
        }
    }
    __syncthreads();
    // ---- re-index the hypotheses (:108-124) ------------------------------------------
    for (int r = wave; r < beam; r += BEAM_NT / 64) {
        const int it = sel_i[r], kb = it / Cm, letter = it % Cm;
        const int hp = h0 + kb, hn = h0 + r;
        const int32_t *src = p.est_in + (size_t)hp * p.Lcap;
        int32_t *dst = p.est_out + (size_t)hn * p.Lcap;
        for (int i = lane; i < p.step; i += 64) dst[i] = src[i];
        if (lane == 0) {
            dst[p.step] = letter;
            p.scores_out[hn] = sel_v[r];
            p.new_input[hn] = letter;
            p.parent[hn] = hp;
        }
    }
    // ---- all finished? (:177-178): count the utterances still below `beam` ------------
    if (tid == 0 && p.finished_count[b] < beam) atomicAdd(p.unfinished, 1);
}

// word 0: done flag, word 1: utterances still searching (scratch), word 2: effective steps
__global__ void beam_done_kernel(int32_t *w) {
    if (w[0]) return;
    w[2] += 1;
    if (w[1] == 0) w[0] = 1;
    w[1] = 0;
}

}  // namespace

extern "C" int asr_tcn_attention_step_f32(const float *eproj, const float *enc,
                                          const int32_t *enc_lens, const float *filt,
                                          const float *glob, const float *w_score,
                                          float b_score, float temperature,
                                          const float *att_prev, const int32_t *parent,
                                          int T, int B, int beam, int A, int Kf, int E,
                                          float *att_new, float *context, void *stream) {
    if (T <= 0 || B <= 0 || beam <= 0 || A <= 0 || E <= 0) return ASR_EINVAL;
    if (Kf != KF) return ASR_EUNSUPPORTED;                  // LocalAttention's default 32 taps
    if (!eproj || !enc || !enc_lens || !filt || !glob || !w_score || !att_prev || !att_new ||
        !context)
        return ASR_EINVAL;
    const size_t lds = (size_t)(KF - 1 + 2 * T + 32) * sizeof(float);
    if (lds > 64 * 1024) return ASR_EUNSUPPORTED;
    AttParams p;
    p.eproj = eproj; p.enc = enc; p.filt = filt; p.glob = glob; p.w_score = w_score;
    p.att_prev = att_prev; p.enc_lens = enc_lens; p.parent = parent;
    p.T = T; p.B = B; p.beam = beam; p.A = A; p.E = E;
    p.b_score = b_score; p.temperature = temperature;
    p.att_new = att_new; p.context = context;
    hipLaunchKernelGGL(tcn_attention_step_kernel, dim3(B * beam), dim3(ATT_NT), lds,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_beam_step_f32(const float *logits, const float *scores_in, float *scores_out,
                                 const int32_t *est_in, int32_t *est_out, int step, int B,
                                 int beam, int C, int Lcap, float len_div,
                                 int32_t *finished_count, float *best_score, int32_t *best_len,
                                 int32_t *best_tokens, int32_t *new_input, int32_t *parent,
                                 int32_t *done_and_scratch, void *stream) {
    if (B <= 0 || beam <= 0 || C < 2 || step < 0 || Lcap <= step) return ASR_EINVAL;
    if (beam > BEAM_MAX || (long)beam * (C - 1) > (long)BEAM_NT * CAND_PER_THREAD)
        return ASR_EUNSUPPORTED;
    if (!logits || !scores_in || !scores_out || !est_in || !est_out || !finished_count ||
        !best_score || !best_len || !best_tokens || !new_input || !parent || !done_and_scratch)
        return ASR_EINVAL;
    BeamParams p;
    p.logits = logits; p.scores_in = scores_in; p.scores_out = scores_out;
    p.est_in = est_in; p.est_out = est_out; p.step = step; p.B = B; p.beam = beam; p.C = C;
    p.Lcap = Lcap; p.len_div = len_div;
    p.finished_count = finished_count; p.best_len = best_len; p.best_tokens = best_tokens;
    p.best_score = best_score; p.new_input = new_input; p.parent = parent;
    p.done = done_and_scratch; p.unfinished = done_and_scratch + 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(BEAM_NT), 0, s, p);
    hipLaunchKernelGGL(beam_done_kernel, dim3(1), dim3(1), 0, s, done_and_scratch);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
