// Step boundary on the device (include/asr_amd.h: asr_grad_sumsq_partials_f32,
// asr_adam_clip_step_f32): the reference's GradientClipping hook
// (modules/hooks/gradient_clipping.py:13-53: clip_grad_norm_, skip the step above
// skip_step_norm) and torch.optim.Adam.step (trainer.py:262-266) as two launches
// that take their decisions from device memory — no read-back between backward and
// the update, so the host runs ahead into the next step's forward.
//
//  1. partial sums of g^2 over the flat gradient bucket (one float per block, plain
//     tree inside a block: deterministic);
//  2. every block of the update adds the partials up in the same order (bit-identical
//     in all blocks), takes norm, clip factor and the skip decision (norm above the
//     threshold, not finite, or the persistent LSTM's error word set) from them and
//     updates its chunk of parameters: m, v (flat, owned by the caller) and the
//     parameter tensors themselves through a chunk table (the parameters stay where
//     torch allocated them).  Block 0 leaves {norm, clipped, skipped, step} for the
//     host to read whenever it likes.
// HBM-bound: 7 floats per parameter element.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 1024;             // elements per table entry / block

__global__ __launch_bounds__(TPB) void sumsq_partials_kernel(const float *__restrict__ g, int64_t n,
                                                             float *__restrict__ partials) {
    __shared__ float scratch[32];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    int64_t hi = lo + per;
    hi = hi > n ? n : hi;
    float s = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += TPB) {
        const float x = g[i];
        s = fmaf(x, x, s);
    }
    s = asr::block_sum(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

struct AdamParams {
    const AsrAdamChunk *chunks;
    const float *g;
    float *m, *v;
    const float *partials;
    int nparts, nchunks;
    const uint32_t *err;
    float lr, beta1, beta2, eps, wd, clip, skip;
    const int32_t *step_in;
    int32_t *step_out;
    float *stats;
};

__global__ __launch_bounds__(TPB) void adam_clip_step_kernel(AdamParams p) {
    __shared__ float scratch[32];
    __shared__ float sh[4];
    // the global norm, the same bits in every block
    float s = 0.f;
    for (int i = threadIdx.x; i < p.nparts; i += TPB) s += p.partials[i];
    s = asr::block_sum(s, scratch);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        const bool err = p.err && *p.err != 0u;
        const bool skip = !(norm <= p.skip) || !isfinite(norm) || err;      // NaN: skip
        const float coef = p.clip / (norm + 1e-6f);                          // clip_grad_norm_'s rule
        const int t = *p.step_in + 1;
        // bias corrections as torch computes them (Python doubles)
        const double b1 = 1.0 - pow((double)p.beta1, (double)t);
        const double b2 = 1.0 - pow((double)p.beta2, (double)t);
        sh[0] = coef < 1.f ? coef : 1.f;
        sh[1] = skip ? 1.f : 0.f;
        sh[2] = (float)((double)p.lr / b1);
        sh[3] = (float)sqrt(b2);
        if (blockIdx.x == 0) {
            p.stats[0] = norm;
            p.stats[1] = coef < 1.f ? 1.f : 0.f;
            p.stats[2] = skip ? 1.f : 0.f;
            p.stats[3] = err ? 1.f : 0.f;
            *p.step_out = skip ? t - 1 : t;
        }
    }
    __syncthreads();
    if (sh[1] != 0.f) return;
    const float coef = sh[0], step_size = sh[2], bc2s = sh[3];
    const float b1 = p.beta1, b2 = p.beta2;
    // (a block walks several chunks: the norm / bias-correction prologue above is paid once)
    for (int ci = blockIdx.x; ci < p.nchunks; ci += gridDim.x) {
        const AsrAdamChunk c = p.chunks[ci];
        float *const w = (float *)c.param;
        for (uint32_t i = threadIdx.x; i < c.count; i += TPB) {
            const size_t f = (size_t)c.flat_offset + i;
            float g = p.g[f] * coef;
            float x = w[i];
            if (p.wd != 0.f) g = fmaf(p.wd, x, g);
            const float m = b1 * p.m[f] + (1.f - b1) * g;           // exp_avg.lerp_(grad, 1 - beta1)
            const float v = b2 * p.v[f] + (1.f - b2) * g * g;       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
            p.m[f] = m;
            p.v[f] = v;
            const float denom = sqrtf(v) / bc2s + p.eps;
            w[i] = x - step_size * (m / denom);
        }
    }
}

// x[t, b, :] *= scale[b], in place; utterances whose factor is exactly 1 are not touched (their
// workgroups leave after one load): the usual case of PathLogSumExp.backward, fst_utils.py:482-485.
__global__ __launch_bounds__(TPB) void scale_rows_kernel(float *x, int T, int B, int C, const float *scale,
                                                         int tchunk) {
    const int b = blockIdx.y;
    const float sc = scale[b];
    if (sc == 1.f) return;
    const int t0 = blockIdx.x * tchunk;
    const int t1 = t0 + tchunk < T ? t0 + tchunk : T;
    for (int t = t0; t < t1; ++t) {
        float *row = x + ((size_t)t * B + b) * C;
        for (int c = threadIdx.x; c < C; c += TPB) row[c] *= sc;
    }
}

}  // namespace

extern "C" int asr_scale_rows_f32(float *x, int T, int B, int C, const float *scale, void *stream) {
    if (T < 0 || B < 0 || C <= 0 || (!x && T > 0 && B > 0) || !scale) return ASR_EINVAL;
    if (T == 0 || B == 0) return ASR_OK;
    if (B > 65535) return ASR_EUNSUPPORTED;
    const int tchunk = 8;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((T + tchunk - 1) / tchunk, B), dim3(TPB), 0,
                       (hipStream_t)stream, x, T, B, C, scale, tchunk);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_adam_chunk_elems(void) { return CHUNK; }

extern "C" int asr_grad_sumsq_partials_f32(const float *g, int64_t n, float *partials, int nparts,
                                           void *stream) {
    if (!g || !partials || n < 0 || nparts <= 0 || nparts > 65535) return ASR_EINVAL;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(nparts), dim3(TPB), 0, (hipStream_t)stream, g, n,
                       partials);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_adam_clip_step_f32(const AsrAdamChunk *chunks, int nchunks, const float *g_flat,
                                      float *m_flat, float *v_flat, const float *partials, int nparts,
                                      const uint32_t *err_word, float lr, float beta1, float beta2,
                                      float eps, float weight_decay, float clip_norm,
                                      float skip_norm, const int32_t *step_in, int32_t *step_out,
                                      float *stats, void *stream) {
    if (!chunks || nchunks <= 0 || !g_flat || !m_flat || !v_flat || !partials || nparts <= 0 ||
        !step_in || !step_out || step_in == step_out || !stats)
        return ASR_EINVAL;
    AdamParams p;
    p.chunks = chunks; p.g = g_flat; p.m = m_flat; p.v = v_flat;
    p.partials = partials; p.nparts = nparts; p.nchunks = nchunks; p.err = err_word;
    p.lr = lr; p.beta1 = beta1; p.beta2 = beta2; p.eps = eps; p.wd = weight_decay;
    p.clip = clip_norm; p.skip = skip_norm;
    p.step_in = step_in; p.step_out = step_out; p.stats = stats;
    hipLaunchKernelGGL(adam_clip_step_kernel, dim3(nchunks < 2048 ? nchunks : 2048), dim3(TPB), 0,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
