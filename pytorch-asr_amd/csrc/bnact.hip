// BatchNorm2d + Hardtanh of the DeepSpeech2 conv front-end, fused (reference
// att_speech/modules/encoders/deep_speech_2.py:60-73: Conv2d -> Normalization
// ('batch_norm', nn.BatchNorm2d) -> Hardtanh(0, 20, inplace)).
//
// HBM-bound streaming over the [B, C, H*W] fp32 convolution output (1.1 GB for
// the first layer at B=512).  torch runs this as: BN statistics, BN normalise,
// clamp, (cast to bf16 for the next convolution) forward and clamp-backward,
// BN dx, BN dscale/dbias backward — about twelve passes over the big tensor.
// Here: one statistics pass + one apply pass forward (the apply writes the
// clamped activation directly in the dtype / layout its consumer wants: bf16
// NCHW for the next convolution, or fp32 time-major [T, B, C, F] for the LSTM
// stack), one reduction pass + one apply pass backward (the clamp mask is
// recomputed from the saved convolution output).
//
// One workgroup per (b, c) plane; per-channel sums are accumulated in double
// with one atomic pair per workgroup.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

using namespace asr;

struct BnParams {
    const float *x;          // [B, C, HW] convolution output
    int B, C, HW, Wd;        // Wd = W (frequency bins); HW = H*W
    const float *gamma, *beta;
    const float *mean, *invstd;   // [C]
    const float *shift;      // [C] or null: the convolution's bias, added on the fly
    float lo, hi;
};

__global__ __launch_bounds__(256) void bn_stats_kernel(const float *x, const float *shift, int C, int HW, double *sums) {
    __shared__ float red[64];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *p = x + ((size_t)b * C + c) * HW;
    const float sh0 = shift ? shift[c] : 0.f;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = p[i] + sh0;
        s += v;
        q += v * v;
    }
    s = block_sum(s, red);
    q = block_sum(q, red + 32);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], (double)s);
        atomicAdd(&sums[2 * c + 1], (double)q);
    }
}

// mean / biased variance -> invstd; running statistics as nn.BatchNorm2d in
// training mode (unbiased variance, momentum)
__global__ void bn_finalize_kernel(const double *sums, int C, double n, float eps, float momentum,
                                   float *mean, float *invstd, float *running_mean,
                                   float *running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / n;
    double var = sums[2 * c + 1] / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// statistics handed over by the producer of x (the convolution's epilogue): sums [2][C] =
// (sum x, sum x^2) of the bias-free input; the folded bias only shifts the mean
__global__ void bn_finalize_from_sums_kernel(const double *sums, const float *shift, int C, double n,
                                             float eps, float momentum, float *mean, float *invstd,
                                             float *running_mean, float *running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mx = sums[c] / n;
    double var = sums[C + c] / n - mx * mx;
    var = var < 0.0 ? 0.0 : var;
    const double m = mx + (shift ? (double)shift[c] : 0.0);
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_stats_kernel(const float *running_mean, const float *running_var, int C,
                                     float eps, float *mean, float *invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// LAYOUT 0: out[b][c][hw];  LAYOUT 1: out[h][b][c][w] (time-major, deep_speech_2.py:142-146)
template <typename OutT, int LAYOUT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnParams p, OutT *out) {
    const int c = blockIdx.x, b = blockIdx.y;
    const float *x = p.x + ((size_t)b * p.C + c) * p.HW;
    const float sc = p.gamma[c] * p.invstd[c];
    const float sh = p.beta[c] - (p.mean[c] - (p.shift ? p.shift[c] : 0.f)) * sc;
    for (int i = threadIdx.x; i < p.HW; i += 256) {
        const float y = fminf(fmaxf(fmaf(x[i], sc, sh), p.lo), p.hi);
        size_t o;
        if (LAYOUT == 0) {
            o = ((size_t)b * p.C + c) * p.HW + i;
        } else {
            const int h = i / p.Wd, w = i - h * p.Wd;
            o = (((size_t)h * p.B + b) * p.C + c) * p.Wd + w;
        }
        out[o] = (OutT)y;
    }
}

template <typename DyT, int LAYOUT>
__device__ __forceinline__ float load_dy(const BnParams &p, const DyT *dy, int b, int c, int i) {
    size_t o;
    if (LAYOUT == 0) {
        o = ((size_t)b * p.C + c) * p.HW + i;
    } else {
        const int h = i / p.Wd, w = i - h * p.Wd;
        o = (((size_t)h * p.B + b) * p.C + c) * p.Wd + w;
    }
    return (float)dy[o];
}

// sums[c] = { sum dyh, sum dyh * xhat } with dyh = dy where lo < bn(x) < hi (hardtanh_backward)
template <typename DyT, int LAYOUT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(BnParams p, const DyT *dy, double *sums) {
    __shared__ float red[64];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *x = p.x + ((size_t)b * p.C + c) * p.HW;
    const float m = p.mean[c] - (p.shift ? p.shift[c] : 0.f), is = p.invstd[c], g = p.gamma[c], be = p.beta[c];
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < p.HW; i += 256) {
        const float xh = (x[i] - m) * is;
        const float y = fmaf(xh, g, be);
        const float d = (y > p.lo && y < p.hi) ? load_dy<DyT, LAYOUT>(p, dy, b, c, i) : 0.f;
        s += d;
        q += d * xh;
    }
    s = block_sum(s, red);
    q = block_sum(q, red + 32);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], (double)s);
        atomicAdd(&sums[2 * c + 1], (double)q);
    }
}

// dx = gamma*invstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)) in training mode,
// gamma*invstd*dyh with running statistics
template <typename DyT, int LAYOUT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BnParams p, const DyT *dy, const double *sums,
                                                               double n, int training, float *dx) {
    const int c = blockIdx.x, b = blockIdx.y;
    const float *x = p.x + ((size_t)b * p.C + c) * p.HW;
    float *o = dx + ((size_t)b * p.C + c) * p.HW;
    const float m = p.mean[c] - (p.shift ? p.shift[c] : 0.f), is = p.invstd[c], g = p.gamma[c], be = p.beta[c];
    const float k1 = training ? (float)(sums[2 * c] / n) : 0.f;
    const float k2 = training ? (float)(sums[2 * c + 1] / n) : 0.f;
    const float gi = g * is;
    for (int i = threadIdx.x; i < p.HW; i += 256) {
        const float xh = (x[i] - m) * is;
        const float y = fmaf(xh, g, be);
        const float d = (y > p.lo && y < p.hi) ? load_dy<DyT, LAYOUT>(p, dy, b, c, i) : 0.f;
        o[i] = gi * (d - k1 - xh * k2);
    }
}


// ---- channels-last input: x[p][c], p = (b*H + h)*W + w.  MIOpen's implicit-GEMM
// convolutions produce and consume this layout, so the big activations never get
// transposed.  A thread owns 4 consecutive channels of a pixel (one 8- or 16-byte
// access; 2-byte accesses per lane reached only 1.7-2.8 TB/s), C/4 threads cover a
// pixel, 1024/C pixels per pass; C % 4 == 0 and C/4 divides 256.
struct BnParamsN {
    const void *x;           // float or __bf16 (template XT), dx has the same type
    int64_t P;               // pixels
    int B, C, H, W;
    const float *gamma, *beta, *mean, *invstd;
    const float *shift;      // [C] or null: the convolution's bias, added on the fly
    float lo, hi;
};

struct F4 { float v[4]; };

__device__ __forceinline__ F4 load4(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    return F4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ F4 load4(const __bf16 *p) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const bf16x4 t = *reinterpret_cast<const bf16x4 *>(p);
    return F4{{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}};
}
__device__ __forceinline__ void store4(float *p, const F4 &a) {
    *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ void store4(__bf16 *p, const F4 &a) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 t;
    t[0] = (__bf16)a.v[0]; t[1] = (__bf16)a.v[1]; t[2] = (__bf16)a.v[2]; t[3] = (__bf16)a.v[3];
    *reinterpret_cast<bf16x4 *>(p) = t;
}

// per-channel totals of a workgroup -> two double atomics per channel
__device__ __forceinline__ void nhwc_block_atomics(const float (&s)[4], const float (&q)[4], int C,
                                                   double *sums) {
    __shared__ float rs[256][4], rq[256][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rs[threadIdx.x][i] = s[i]; rq[threadIdx.x][i] = q[i]; }
    __syncthreads();
    const int CG = C / 4;
    if ((int)threadIdx.x < C) {
        const int cg = threadIdx.x >> 2, i = threadIdx.x & 3;
        float a = 0.f, b = 0.f;
        for (int t = cg; t < 256; t += CG) { a += rs[t][i]; b += rq[t][i]; }
        atomicAdd(&sums[2 * threadIdx.x], (double)a);
        atomicAdd(&sums[2 * threadIdx.x + 1], (double)b);
    }
}

#define NHWC_THREAD_SETUP(P_)                                                          \
    const int CG = C / 4, PL = 256 / CG, c0 = 4 * ((int)threadIdx.x % CG),             \
              pl = (int)threadIdx.x / CG;                                              \
    const int64_t per = ((P_) + gridDim.x - 1) / gridDim.x;                            \
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < (P_) ? p0 + per : (P_)

template <typename XT>
__global__ __launch_bounds__(256) void bn_stats_nhwc_kernel(const XT *x, const float *shift, int64_t P, int C, double *sums) {
    NHWC_THREAD_SETUP(P);
    float sh0[4], s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) sh0[i] = shift ? shift[c0 + i] : 0.f;
    // four pixels per trip, all four loads issued before the first add: one 8-byte load in
    // flight per thread kept the pass at 2.4 TB/s (latency-bound); same accumulation order
    int64_t p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
        F4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load4(x + (p + u * PL) * C + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = v[u].v[i] + sh0[i];
                s[i] += t;
                q[i] += t * t;
            }
    }
    for (; p < p1; p += PL) {
        const F4 v = load4(x + p * C + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u = v.v[i] + sh0[i];
            s[i] += u;
            q[i] += u * u;
        }
    }
    nhwc_block_atomics(s, q, C, sums);
}

// time-major output / dy index of pixel p, channel c: [H][B][C][W]
__device__ __forceinline__ size_t tm_index(const BnParamsN &p, int64_t pix, int c) {
    const int w = (int)(pix % p.W);
    const int64_t r = pix / p.W;
    const int h = (int)(r % p.H), b = (int)(r / p.H);
    return (((size_t)h * p.B + b) * p.C + c) * p.W + w;
}

template <typename XT, typename OutT, int TM>
__global__ __launch_bounds__(256) void bn_act_fwd_nhwc_kernel(BnParamsN p, OutT *out) {
    const XT *px = (const XT *)p.x;
    const int C = p.C;
    NHWC_THREAD_SETUP(p.P);
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = p.gamma[c0 + i] * p.invstd[c0 + i];
        sh[i] = p.beta[c0 + i] - (p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f)) * sc[i];
    }
    for (int64_t pix = p0 + pl; pix < p1; pix += PL) {
        const F4 v = load4(px + pix * C + c0);
        F4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) y.v[i] = fminf(fmaxf(fmaf(v.v[i], sc[i], sh[i]), p.lo), p.hi);
        if (TM) {
            const size_t o = tm_index(p, pix, c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) out[o + (size_t)i * p.W] = (OutT)y.v[i];
        } else {
            store4(out + pix * C + c0, y);
        }
    }
}

template <typename DyT, int TM>
__device__ __forceinline__ F4 load_dy4(const BnParamsN &p, const DyT *dy, int64_t pix, int c0) {
    if (TM) {
        const size_t o = tm_index(p, pix, c0);
        F4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) d.v[i] = (float)dy[o + (size_t)i * p.W];
        return d;
    }
    return load4(dy + pix * p.C + c0);
}

template <typename XT, typename DyT, int TM>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_nhwc_kernel(BnParamsN p, const DyT *dy, double *sums) {
    const XT *px = (const XT *)p.x;
    const int C = p.C;
    NHWC_THREAD_SETUP(p.P);
    float m[4], is[4], g[4], be[4], s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f);
        is[i] = p.invstd[c0 + i]; g[i] = p.gamma[c0 + i]; be[i] = p.beta[c0 + i];
    }
    auto acc = [&](const F4 &v, const F4 &dv) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v.v[i] - m[i]) * is[i];
            const float y = fmaf(xh, g[i], be[i]);
            const float d = (y > p.lo && y < p.hi) ? dv.v[i] : 0.f;
            s[i] += d;
            q[i] += d * xh;
        }
    };
    int64_t pix = p0 + pl;
    for (; pix + PL < p1; pix += 2 * PL) {        // two pixels per trip: four loads in flight
        const F4 v0 = load4(px + pix * C + c0), v1 = load4(px + (pix + PL) * C + c0);
        const F4 d0 = load_dy4<DyT, TM>(p, dy, pix, c0), d1 = load_dy4<DyT, TM>(p, dy, pix + PL, c0);
        acc(v0, d0);
        acc(v1, d1);
    }
    for (; pix < p1; pix += PL) acc(load4(px + pix * C + c0), load_dy4<DyT, TM>(p, dy, pix, c0));
    nhwc_block_atomics(s, q, C, sums);
}

template <typename XT, typename DyT, int TM>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_nhwc_kernel(BnParamsN p, const DyT *dy, const double *sums,
                                                                    double n, int training, XT *dx) {
    const XT *px = (const XT *)p.x;
    const int C = p.C;
    NHWC_THREAD_SETUP(p.P);
    float m[4], is[4], g[4], be[4], k1[4], k2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f);
        is[i] = p.invstd[c0 + i]; g[i] = p.gamma[c0 + i]; be[i] = p.beta[c0 + i];
        k1[i] = training ? (float)(sums[2 * (c0 + i)] / n) : 0.f;
        k2[i] = training ? (float)(sums[2 * (c0 + i) + 1] / n) : 0.f;
    }
    for (int64_t pix = p0 + pl; pix < p1; pix += PL) {
        const F4 v = load4(px + pix * C + c0);
        const F4 dv = load_dy4<DyT, TM>(p, dy, pix, c0);
        F4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v.v[i] - m[i]) * is[i];
            const float y = fmaf(xh, g[i], be[i]);
            const float d = (y > p.lo && y < p.hi) ? dv.v[i] : 0.f;
            o.v[i] = g[i] * is[i] * (d - k1[i] - xh * k2[i]);
        }
        store4(dx + pix * C + c0, o);
    }
}

// ---- the same three passes with EIGHT channels per thread (16-byte accesses on bf16
// tensors; with four, i.e. 8 bytes per lane, the bf16 passes ran at 2.3-2.7 TB/s); plain
// channels-last on both sides
struct F8 { float v[8]; };
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
__device__ __forceinline__ F8 load8(const __bf16 *p) {
    const bf16x8v t = *reinterpret_cast<const bf16x8v *>(p);
    F8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)t[i];
    return r;
}
__device__ __forceinline__ F8 load8(const float *p) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    return F8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void store8(__bf16 *p, const F8 &a) {
    bf16x8v t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (__bf16)a.v[i];
    *reinterpret_cast<bf16x8v *>(p) = t;
}
__device__ __forceinline__ void store8(float *p, const F8 &a) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// per-channel totals of a workgroup (8 channels per thread) -> two double atomics per channel
__device__ __forceinline__ void nhwc_block_atomics8(const float (&s)[8], const float (&q)[8], int C,
                                                    double *sums) {
    __shared__ float rs8[256][9], rq8[256][9];
#pragma unroll
    for (int i = 0; i < 8; ++i) { rs8[threadIdx.x][i] = s[i]; rq8[threadIdx.x][i] = q[i]; }
    __syncthreads();
    const int CG = C / 8;
    if ((int)threadIdx.x < C) {
        const int cg = threadIdx.x >> 3, i = threadIdx.x & 7;
        float a = 0.f, b = 0.f;
        for (int t = cg; t < 256; t += CG) { a += rs8[t][i]; b += rq8[t][i]; }
        atomicAdd(&sums[2 * threadIdx.x], (double)a);
        atomicAdd(&sums[2 * threadIdx.x + 1], (double)b);
    }
}
// Work distribution: chunks of 2*PL pixels, grid-strided — the workgroups running at any time
// then cover one contiguous window of the tensor.  With one private contiguous slice per
// workgroup (as in the 4-channel kernels above) 2048 concurrent workgroups stream 2048
// regions 76 KB apart and the bf16 passes ran at 2.4 TB/s of read + write.
#define NHWC8_THREAD_SETUP(P_)                                                         \
    const int CG = C / 8, PL = 256 / CG, c0 = 8 * ((int)threadIdx.x % CG),             \
              pl = (int)threadIdx.x / CG;                                              \
    const int64_t pend = (P_), cstep = (int64_t)gridDim.x * 2 * PL

template <typename XT, typename OutT>
__global__ __launch_bounds__(256) void bn_act_fwd_nhwc8_kernel(BnParamsN p, OutT *out) {
    const XT *px = (const XT *)p.x;
    const int C = p.C;
    NHWC8_THREAD_SETUP(p.P);
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = p.gamma[c0 + i] * p.invstd[c0 + i];
        sh[i] = p.beta[c0 + i] - (p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f)) * sc[i];
    }
    auto one = [&](int64_t pix, const F8 &v) {
        F8 y;
#pragma unroll
        for (int i = 0; i < 8; ++i) y.v[i] = fminf(fmaxf(fmaf(v.v[i], sc[i], sh[i]), p.lo), p.hi);
        store8(out + pix * C + c0, y);
    };
    for (int64_t pix = (int64_t)blockIdx.x * 2 * PL + pl; pix < pend; pix += cstep) {
        if (pix + PL < pend) {
            const F8 v0 = load8(px + pix * C + c0), v1 = load8(px + (pix + PL) * C + c0);
            one(pix, v0);
            one(pix + PL, v1);
        } else {
            one(pix, load8(px + pix * C + c0));
        }
    }
}

// MODE 0: sums[c] = { sum dyh, sum dyh * xhat };  MODE 1: dx
template <typename XT, typename DyT, int MODE>
__global__ __launch_bounds__(256) void bn_act_bwd_nhwc8_kernel(BnParamsN p, const DyT *dy, double *sums,
                                                               double n, int training, XT *dx) {
    const XT *px = (const XT *)p.x;
    const int C = p.C;
    NHWC8_THREAD_SETUP(p.P);
    float m[8], is[8], g[8], be[8], k1[8], k2[8], s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        m[i] = p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f);
        is[i] = p.invstd[c0 + i]; g[i] = p.gamma[c0 + i]; be[i] = p.beta[c0 + i];
        k1[i] = MODE == 1 && training ? (float)(sums[2 * (c0 + i)] / n) : 0.f;
        k2[i] = MODE == 1 && training ? (float)(sums[2 * (c0 + i) + 1] / n) : 0.f;
        s[i] = q[i] = 0.f;
    }
    auto one = [&](int64_t pix, const F8 &v, const F8 &dv) {
        F8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (v.v[i] - m[i]) * is[i];
            const float y = fmaf(xh, g[i], be[i]);
            const float d = (y > p.lo && y < p.hi) ? dv.v[i] : 0.f;
            if (MODE == 0) { s[i] += d; q[i] += d * xh; }
            else o.v[i] = g[i] * is[i] * (d - k1[i] - xh * k2[i]);
        }
        if (MODE == 1) store8(dx + pix * C + c0, o);
    };
    for (int64_t pix = (int64_t)blockIdx.x * 2 * PL + pl; pix < pend; pix += cstep) {
        if (pix + PL < pend) {                    // two pixels per trip: four loads in flight
            const F8 v0 = load8(px + pix * C + c0), v1 = load8(px + (pix + PL) * C + c0);
            const F8 d0 = load8(dy + pix * C + c0), d1 = load8(dy + (pix + PL) * C + c0);
            one(pix, v0, d0);
            one(pix + PL, v1, d1);
        } else {
            one(pix, load8(px + pix * C + c0), load8(dy + pix * C + c0));
        }
    }
    if (MODE == 0) nhwc_block_atomics8(s, q, C, sums);
}

// ---- channels-last x, TIME-MAJOR y / dy ([H][B][C][W], the LSTM stack's layout): per
// (b, h) row the W x C block of x is the transpose of the C x W block of y.  The kernels
// above with TM = 1 address y element-wise (4-byte accesses W elements apart: 2.2 TB/s);
// these move rows of y / dy as whole contiguous 16- or 8-byte pieces and transpose through
// an fp32 LDS tile of TMR rows (bank pattern: 4 W cg + w over a half-wave = 32 distinct
// banks for W odd).  Same arithmetic, same accumulation order per thread.
// x / d for x < 2^20, d < 2^10 by multiplication (m = 2^32 / d + 1; d = 1: x itself).  The tile
// loops below used run-time `/`: 3-4 integer divisions (one of them 64-bit) per 8- or 16-byte
// piece — 40-100 VALU instructions each — held these kernels at 2.5 TB/s.
struct FastDiv {
    unsigned m, d;
    __device__ __forceinline__ explicit FastDiv(int dd) : m((unsigned)((1ull << 32) / (unsigned)dd) + 1u), d((unsigned)dd) {}
    __device__ __forceinline__ int operator()(int x) const { return d == 1u ? x : (int)__umulhi((unsigned)x, m); }
};

template <typename T>
__device__ __forceinline__ void tm_row_to_lds(const T *row, float *tile, int CW, int it) {
    const F4 v = load4(row + 4 * it);
    *reinterpret_cast<float4 *>(tile + 4 * it) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
}

template <typename XT, typename OutT>
__global__ __launch_bounds__(256) void bn_act_fwd_nhwc_tm_kernel(BnParamsN p, OutT *out, int TMR) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const XT *px = (const XT *)p.x;
    const int C = p.C, W = p.W, CW = C * W, CG = C / 4, Q = CW / 4;
    const int c0 = 4 * ((int)threadIdx.x % CG);
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = p.gamma[c0 + i] * p.invstd[c0 + i];
        sh[i] = p.beta[c0 + i] - (p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f)) * sc[i];
    }
    const int64_t nrows = p.P / W;
    const FastDiv dCG(CG), dW(W), dQ(Q);
    for (int64_t r0 = (int64_t)blockIdx.x * TMR; r0 < nrows; r0 += (int64_t)gridDim.x * TMR) {
        const int nr = nrows - r0 < TMR ? (int)(nrows - r0) : TMR;
        const int b0 = (int)(r0 / p.H), h0 = (int)(r0 - (int64_t)b0 * p.H);     // (uniform: once per tile)
        for (int it = threadIdx.x; it < nr * W * CG; it += 256) {
            const int j = dCG(it), row = dW(j), w = j - row * W;      // pixel j of the tile
            const F4 v = load4(px + (r0 * W + j) * C + c0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                tile[row * CW + (c0 + i) * W + w] = fminf(fmaxf(fmaf(v.v[i], sc[i], sh[i]), p.lo), p.hi);
        }
        __syncthreads();
        for (int it = threadIdx.x; it < nr * Q; it += 256) {
            const int row = dQ(it), k = it - row * Q;
            int h = h0 + row, b = b0;
            while (h >= p.H) { h -= p.H; ++b; }                       // (a tile spans few rows)
            const float4 t = *reinterpret_cast<const float4 *>(tile + row * CW + 4 * k);
            store4(out + ((size_t)h * p.B + b) * CW + 4 * k, F4{{t.x, t.y, t.z, t.w}});
        }
        __syncthreads();
    }
}

// MODE 0: sums[c] = { sum dyh, sum dyh * xhat };  MODE 1: dx
template <typename XT, typename DyT, int MODE>
__global__ __launch_bounds__(256) void bn_act_bwd_nhwc_tm_kernel(BnParamsN p, const DyT *dy, double *sums,
                                                                 double n, int training, XT *dx, int TMR) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const XT *px = (const XT *)p.x;
    const int C = p.C, W = p.W, CW = C * W, CG = C / 4, Q = CW / 4;
    const int c0 = 4 * ((int)threadIdx.x % CG);
    float m[4], is[4], g[4], be[4], k1[4], k2[4], s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = p.mean[c0 + i] - (p.shift ? p.shift[c0 + i] : 0.f);
        is[i] = p.invstd[c0 + i]; g[i] = p.gamma[c0 + i]; be[i] = p.beta[c0 + i];
        k1[i] = MODE == 1 && training ? (float)(sums[2 * (c0 + i)] / n) : 0.f;
        k2[i] = MODE == 1 && training ? (float)(sums[2 * (c0 + i) + 1] / n) : 0.f;
    }
    const int64_t nrows = p.P / W;
    const FastDiv dCG(CG), dW(W), dQ(Q);
    for (int64_t r0 = (int64_t)blockIdx.x * TMR; r0 < nrows; r0 += (int64_t)gridDim.x * TMR) {
        const int nr = nrows - r0 < TMR ? (int)(nrows - r0) : TMR;
        const int b0 = (int)(r0 / p.H), h0 = (int)(r0 - (int64_t)b0 * p.H);     // (uniform: once per tile)
        for (int it = threadIdx.x; it < nr * Q; it += 256) {
            const int row = dQ(it), k = it - row * Q;
            int h = h0 + row, b = b0;
            while (h >= p.H) { h -= p.H; ++b; }                       // (a tile spans few rows)
            tm_row_to_lds(dy + ((size_t)h * p.B + b) * CW, tile + row * CW, CW, k);
        }
        __syncthreads();
        for (int it = threadIdx.x; it < nr * W * CG; it += 256) {
            const int j = dCG(it), row = dW(j), w = j - row * W;
            const F4 v = load4(px + (r0 * W + j) * C + c0);
            F4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (v.v[i] - m[i]) * is[i];
                const float y = fmaf(xh, g[i], be[i]);
                const float d = (y > p.lo && y < p.hi) ? tile[row * CW + (c0 + i) * W + w] : 0.f;
                if (MODE == 0) { s[i] += d; q[i] += d * xh; }
                else o.v[i] = g[i] * is[i] * (d - k1[i] - xh * k2[i]);
            }
            if (MODE == 1) store4(dx + (r0 * W + j) * C + c0, o);
        }
        __syncthreads();
    }
    if (MODE == 0) nhwc_block_atomics(s, q, C, sums);
}

// dgamma, dbeta, and the gradient of the folded-in convolution bias: sum of dx over
// all pixels = gamma*invstd*sum(dyh) with running statistics, exactly 0 with batch
// statistics (sum of xhat is 0 by construction)
__global__ void bn_param_grads_kernel(const double *sums, int C, const float *gamma,
                                      const float *invstd, int training, float *dgamma,
                                      float *dbeta, float *dshift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums[2 * c];
    dgamma[c] = (float)sums[2 * c + 1];
    if (dshift) dshift[c] = training ? 0.f : gamma[c] * invstd[c] * (float)sums[2 * c];
}

__global__ void zero_doubles_kernel(double *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// rows of a time-major tile (fp32, cw floats per row).  8, not as many as fit: with 32 rows
// (45 KB of LDS at cw = 352) three workgroups shared a CU and the load -> transpose -> store
// phases of a tile had nothing to overlap with: 130 / 216 / 214 us forward / backward reduce /
// backward apply at B=768; at 8 rows 74 / 169 / 111 (16 rows: 95 / 186 / 157, 4: 90 / 188 / 137).
inline int tm_rows(int cw) {
    if (cw <= 0) return 0;
    const int fit = 48 * 1024 / (cw * 4);
    return fit < 8 ? fit : 8;
}

inline bool bad_shape(int B, int C, int H, int W) {
    return B <= 0 || C <= 0 || H <= 0 || W <= 0 || C > 65535 || B > 65535 ||
           (int64_t)H * W > 0x7fffffff;
}

}  // namespace

extern "C" int64_t asr_bn_act_workspace_bytes(int C) { return C < 0 ? -1 : (int64_t)C * 2 * 8 + 64; }

extern "C" int asr_bn_act_fwd_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                                  const float *gamma, const float *beta,
                                  float *running_mean, float *running_var,
                                  int channels_last,
                                  int training, float momentum, float eps, float lo, float hi,
                                  void *out, int out_bf16, int out_time_major,
                                  float *save_mean, float *save_invstd,
                                  const double *chan_sums,
                                  void *workspace, int64_t workspace_bytes, void *stream) {
    if (bad_shape(B, C, H, W) || !x || !gamma || !beta || !out || !save_mean || !save_invstd)
        return ASR_EINVAL;
    if (!training && (!running_mean || !running_var)) return ASR_EINVAL;
    if (!workspace || workspace_bytes < asr_bn_act_workspace_bytes(C)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double *sums = (double *)workspace;
    const int HW = H * W;
    const dim3 grid(C, B);
    if (channels_last && (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0)) return ASR_EUNSUPPORTED;
    if (x_bf16 && !channels_last) return ASR_EUNSUPPORTED;   // bf16 input: channels-last only
    const int64_t P = (int64_t)B * HW;
    const int nwg = (int)(P / 64 < 4096 ? (P / 64 > 0 ? P / 64 : 1) : 4096);
    if (training && chan_sums) {
        hipLaunchKernelGGL(bn_finalize_from_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, s, chan_sums,
                           conv_bias, C, (double)B * HW, eps, momentum, save_mean, save_invstd,
                           running_mean, running_var);
    } else if (training) {
        hipLaunchKernelGGL(zero_doubles_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, sums, 2 * C);
        if (channels_last)
            if (x_bf16) hipLaunchKernelGGL(bn_stats_nhwc_kernel<__bf16>, dim3(nwg), dim3(256), 0, s, (const __bf16 *)x, conv_bias, P, C, sums);
            else hipLaunchKernelGGL(bn_stats_nhwc_kernel<float>, dim3(nwg), dim3(256), 0, s, (const float *)x, conv_bias, P, C, sums);
        else
            hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, (const float *)x, conv_bias, C, HW, sums);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, C,
                           (double)B * HW, eps, momentum, save_mean, save_invstd, running_mean,
                           running_var);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, s, running_mean,
                           running_var, C, eps, save_mean, save_invstd);
    }
    BnParams p;
    p.x = (const float *)x; p.B = B; p.C = C; p.HW = HW; p.Wd = W; p.gamma = gamma; p.beta = beta;
    p.mean = save_mean; p.invstd = save_invstd; p.shift = conv_bias; p.lo = lo; p.hi = hi;
    if (channels_last) {
        BnParamsN q;
        q.x = x; q.P = P; q.B = B; q.C = C; q.H = H; q.W = W; q.gamma = gamma; q.beta = beta;
        q.mean = save_mean; q.invstd = save_invstd; q.shift = conv_bias; q.lo = lo; q.hi = hi;
        // time-major output through the LDS transpose when a row of it is whole 16-byte pieces
        const int cw = C * W, tmr = tm_rows(cw);
        if (out_time_major && cw % 8 == 0 && tmr >= 1) {
            const int64_t rows = (int64_t)B * H;
            const int g2 = (int)((rows + tmr - 1) / tmr < 2048 ? (rows + tmr - 1) / tmr : 2048);
            const size_t lds = (size_t)tmr * cw * 4;
#define ASR_BN_FWDT(XT, OT) \
            hipLaunchKernelGGL((bn_act_fwd_nhwc_tm_kernel<XT, OT>), dim3(g2), dim3(256), lds, s, q, (OT *)out, tmr)
            if (x_bf16) { if (out_bf16) ASR_BN_FWDT(__bf16, __bf16); else ASR_BN_FWDT(__bf16, float); }
            else { if (out_bf16) ASR_BN_FWDT(float, __bf16); else ASR_BN_FWDT(float, float); }
#undef ASR_BN_FWDT
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
        }
        if (!out_time_major && x_bf16 && C % 8 == 0 && 256 % (C / 8) == 0) {      // 16-byte accesses
            if (out_bf16) hipLaunchKernelGGL((bn_act_fwd_nhwc8_kernel<__bf16, __bf16>), dim3(nwg), dim3(256), 0, s, q, (__bf16 *)out);
            else hipLaunchKernelGGL((bn_act_fwd_nhwc8_kernel<__bf16, float>), dim3(nwg), dim3(256), 0, s, q, (float *)out);
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
        }
#define ASR_BN_FWDN(XT, OT, TMV) \
        hipLaunchKernelGGL((bn_act_fwd_nhwc_kernel<XT, OT, TMV>), dim3(nwg), dim3(256), 0, s, q, (OT *)out)
        if (x_bf16) {
            if (out_bf16) { if (out_time_major) ASR_BN_FWDN(__bf16, __bf16, 1); else ASR_BN_FWDN(__bf16, __bf16, 0); }
            else { if (out_time_major) ASR_BN_FWDN(__bf16, float, 1); else ASR_BN_FWDN(__bf16, float, 0); }
        } else {
            if (out_bf16) { if (out_time_major) ASR_BN_FWDN(float, __bf16, 1); else ASR_BN_FWDN(float, __bf16, 0); }
            else { if (out_time_major) ASR_BN_FWDN(float, float, 1); else ASR_BN_FWDN(float, float, 0); }
        }
#undef ASR_BN_FWDN
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    if (out_bf16) {
        if (out_time_major) hipLaunchKernelGGL((bn_act_fwd_kernel<__bf16, 1>), grid, dim3(256), 0, s, p, (__bf16 *)out);
        else hipLaunchKernelGGL((bn_act_fwd_kernel<__bf16, 0>), grid, dim3(256), 0, s, p, (__bf16 *)out);
    } else {
        if (out_time_major) hipLaunchKernelGGL((bn_act_fwd_kernel<float, 1>), grid, dim3(256), 0, s, p, (float *)out);
        else hipLaunchKernelGGL((bn_act_fwd_kernel<float, 0>), grid, dim3(256), 0, s, p, (float *)out);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// phase 0: everything; 1: the per-channel sums (left in the workspace) and the parameter
// gradients, which are local quantities; 2: dx from the sums the caller left in the workspace
// (e.g. all-reduced over the replicas) and the element count n_total they stand for.
extern "C" int asr_bn_act_bwd_phase_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                                  const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd,
                                  int channels_last,
                                  int training, float lo, float hi,
                                  const void *dy, int dy_bf16, int dy_time_major,
                                  void *dx, float *dgamma, float *dbeta, float *dconv_bias,
                                  void *workspace, int64_t workspace_bytes, int phase,
                                  double n_total, void *stream) {
    if (phase < 0 || phase > 2 || (phase == 2 && !(n_total >= 1.0))) return ASR_EINVAL;
    const bool do_reduce = phase != 2, do_apply = phase != 1;
    if (bad_shape(B, C, H, W) || !x || !gamma || !beta || !save_mean || !save_invstd || !dy ||
        !dx || !dgamma || !dbeta)
        return ASR_EINVAL;
    if (!workspace || workspace_bytes < asr_bn_act_workspace_bytes(C)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double *sums = (double *)workspace;
    const int HW = H * W;
    const dim3 grid(C, B);
    BnParams p;
    p.x = (const float *)x; p.B = B; p.C = C; p.HW = HW; p.Wd = W; p.gamma = gamma; p.beta = beta;
    p.mean = save_mean; p.invstd = save_invstd; p.shift = conv_bias; p.lo = lo; p.hi = hi;
    const double n = phase == 2 ? n_total : (double)B * HW;
    if (do_reduce)
        hipLaunchKernelGGL(zero_doubles_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, sums, 2 * C);
    if (x_bf16 && !channels_last) return ASR_EUNSUPPORTED;
    if (channels_last) {
        if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return ASR_EUNSUPPORTED;
        BnParamsN q;
        q.x = x; q.P = (int64_t)B * HW; q.B = B; q.C = C; q.H = H; q.W = W; q.gamma = gamma;
        q.beta = beta; q.mean = save_mean; q.invstd = save_invstd; q.shift = conv_bias; q.lo = lo; q.hi = hi;
        const int nwg = (int)(q.P / 64 < 4096 ? (q.P / 64 > 0 ? q.P / 64 : 1) : 4096);
        const int cw = C * W, tmr = tm_rows(cw);
        if (dy_time_major && cw % 8 == 0 && tmr >= 1) {
            const int64_t rows = (int64_t)B * H;
            const int g2 = (int)((rows + tmr - 1) / tmr < 2048 ? (rows + tmr - 1) / tmr : 2048);
            const size_t lds = (size_t)tmr * cw * 4;
#define ASR_BN_BWDT(XT, DT)                                                                          \
            do {                                                                                     \
                if (do_reduce)                                                                       \
                hipLaunchKernelGGL((bn_act_bwd_nhwc_tm_kernel<XT, DT, 0>), dim3(g2), dim3(256), lds, \
                                   s, q, (const DT *)dy, sums, n, training, (XT *)dx, tmr);          \
                if (do_apply)                                                                        \
                hipLaunchKernelGGL((bn_act_bwd_nhwc_tm_kernel<XT, DT, 1>), dim3(g2), dim3(256), lds, \
                                   s, q, (const DT *)dy, sums, n, training, (XT *)dx, tmr);          \
            } while (0)
            if (x_bf16) { if (dy_bf16) ASR_BN_BWDT(__bf16, __bf16); else ASR_BN_BWDT(__bf16, float); }
            else { if (dy_bf16) ASR_BN_BWDT(float, __bf16); else ASR_BN_BWDT(float, float); }
#undef ASR_BN_BWDT
            if (do_reduce) hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, C, gamma, save_invstd, training, dgamma, dbeta, dconv_bias);
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
        }
        if (!dy_time_major && x_bf16 && C % 8 == 0 && 256 % (C / 8) == 0) {       // 16-byte accesses
#define ASR_BN_BWD8(DT)                                                                            \
            do {                                                                                   \
                if (do_reduce)                                                                     \
                hipLaunchKernelGGL((bn_act_bwd_nhwc8_kernel<__bf16, DT, 0>), dim3(nwg), dim3(256), 0, s, \
                                   q, (const DT *)dy, sums, n, training, (__bf16 *)dx);            \
                if (do_apply)                                                                      \
                hipLaunchKernelGGL((bn_act_bwd_nhwc8_kernel<__bf16, DT, 1>), dim3(nwg), dim3(256), 0, s, \
                                   q, (const DT *)dy, sums, n, training, (__bf16 *)dx);            \
            } while (0)
            if (dy_bf16) ASR_BN_BWD8(__bf16); else ASR_BN_BWD8(float);
#undef ASR_BN_BWD8
            if (do_reduce) hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, C, gamma, save_invstd, training, dgamma, dbeta, dconv_bias);
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
        }
#define ASR_BN_BWDN(XT, DT, TMV)                                                                  \
        do {                                                                                      \
            if (do_reduce)                                                                        \
            hipLaunchKernelGGL((bn_act_bwd_reduce_nhwc_kernel<XT, DT, TMV>), dim3(nwg), dim3(256), \
                               0, s, q, (const DT *)dy, sums);                                    \
            if (do_apply)                                                                         \
            hipLaunchKernelGGL((bn_act_bwd_apply_nhwc_kernel<XT, DT, TMV>), dim3(nwg), dim3(256),  \
                               0, s, q, (const DT *)dy, sums, n, training, (XT *)dx);             \
        } while (0)
        if (x_bf16) {
            if (dy_bf16) { if (dy_time_major) ASR_BN_BWDN(__bf16, __bf16, 1); else ASR_BN_BWDN(__bf16, __bf16, 0); }
            else { if (dy_time_major) ASR_BN_BWDN(__bf16, float, 1); else ASR_BN_BWDN(__bf16, float, 0); }
        } else {
            if (dy_bf16) { if (dy_time_major) ASR_BN_BWDN(float, __bf16, 1); else ASR_BN_BWDN(float, __bf16, 0); }
            else { if (dy_time_major) ASR_BN_BWDN(float, float, 1); else ASR_BN_BWDN(float, float, 0); }
        }
#undef ASR_BN_BWDN
        if (do_reduce) hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, C, gamma, save_invstd, training, dgamma, dbeta, dconv_bias);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
#define ASR_BN_BWD(DT, LAY)                                                                       \
    do {                                                                                          \
        if (do_reduce)                                                                            \
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<DT, LAY>), grid, dim3(256), 0, s, p,         \
                           (const DT *)dy, sums);                                                 \
        if (do_apply)                                                                             \
        hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DT, LAY>), grid, dim3(256), 0, s, p,          \
                           (const DT *)dy, sums, n, training, (float *)dx);                       \
    } while (0)
    if (dy_bf16) {
        if (dy_time_major) ASR_BN_BWD(__bf16, 1); else ASR_BN_BWD(__bf16, 0);
    } else {
        if (dy_time_major) ASR_BN_BWD(float, 1); else ASR_BN_BWD(float, 0);
    }
#undef ASR_BN_BWD
    if (do_reduce) hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, C, gamma, save_invstd, training, dgamma, dbeta, dconv_bias);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_bn_act_bwd_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                                  const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd,
                                  int channels_last,
                                  int training, float lo, float hi,
                                  const void *dy, int dy_bf16, int dy_time_major,
                                  void *dx, float *dgamma, float *dbeta, float *dconv_bias,
                                  void *workspace, int64_t workspace_bytes, void *stream) {
    return asr_bn_act_bwd_phase_f32(x, x_bf16, conv_bias, B, C, H, W, gamma, beta, save_mean, save_invstd,
                                    channels_last, training, lo, hi, dy, dy_bf16, dy_time_major, dx, dgamma,
                                    dbeta, dconv_bias, workspace, workspace_bytes, 0, 0.0, stream);
}
