// Row-group log-softmax (forward / backward) and the per-frame max
// stabilisation of FSTDecoder.get_fst_loss for gfx950.
// Reference arithmetic: att_speech/modules/ctc_losses.py:29-43
// (get_normalized_acts) and modules/decoders/advanced_decoder.py:479-484.
//
// HBM-bound streaming kernels: one 64-lane wave owns one row-group, lanes read
// consecutive floats (coalesced 256 B per wave instruction), the group is kept
// in registers between the reduction and the write, so every element is read
// once and written once.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

using namespace asr;

// PER = ceil(group / 64) elements per lane, compile-time so the row stays in
// registers.
template <int PER>
__global__ void log_softmax_fwd_kernel(const float *__restrict__ x,
                                       float *__restrict__ y, int64_t rows,
                                       int group) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *xr = x + r * group;
        float v[PER];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            v[i] = c < group ? xr[c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) s += __expf(v[i] - m);
        s = wave_sum(s);
        const float l = m + __logf(s);
        float *yr = y + r * group;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            if (c < group) yr[c] = v[i] - l;
        }
    }
}

template <int PER>
__global__ void log_softmax_bwd_kernel(const float *__restrict__ y,
                                       const float *__restrict__ dy,
                                       float *__restrict__ dx, int64_t rows,
                                       int group) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *yr = y + r * group;
        const float *gr = dy + r * group;
        float vy[PER], vg[PER];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            vy[i] = c < group ? yr[c] : -INFINITY;
            vg[i] = c < group ? gr[c] : 0.f;
            s += vg[i];
        }
        s = wave_sum(s);
        float *dr = dx + r * group;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            if (c < group) dr[c] = vg[i] - __expf(vy[i]) * s;
        }
    }
}

// one wave per (t,b) row; the masked per-utterance sum of maxima is reduced
// by max_sum_kernel in a fixed order (bitwise reproducible, no float atomics).
template <int PER>
__global__ void sub_rowmax_kernel(const float *__restrict__ x,
                                  float *__restrict__ y,
                                  float *__restrict__ row_max, int T, int B,
                                  int C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t rows = (int64_t)T * B;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *xr = x + r * C;
        float v[PER];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            v[i] = c < C ? xr[c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float *yr = y + r * C;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            if (c < C) yr[c] = v[i] - m;
        }
        if (lane == 0) row_max[r] = m;
    }
}

// max_sum[b] = sum_{t < lens[b]} row_max[t,b]; one wave per utterance.
__global__ void max_sum_kernel(const float *__restrict__ row_max,
                               const int32_t *__restrict__ lens,
                               float *__restrict__ max_sum, int T, int B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int len = lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    float s = 0.f;
    for (int t = lane; t < len; t += 64) s += row_max[(size_t)t * B + b];
    s = wave_sum(s);
    if (lane == 0) max_sum[b] = s;
}

// first maximum of every row (one wave per row): CTCDecoderAdvanced.decode's
// per-frame arg-max (advanced_decoder.py:352).
template <int PER>
__global__ void argmax_rows_kernel(const float *__restrict__ x,
                                   int32_t *__restrict__ out, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *xr = x + r * C;
        float best = -INFINITY;
        int arg = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int c = i * 64 + lane;
            float v = c < C ? xr[c] : -INFINITY;
            if (c < C && (v > best || arg == 0x7fffffff)) { best = v; arg = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (lane == 0) out[r] = arg;
    }
}

inline int grid_for(int64_t rows) {
    int64_t blocks = (rows + 3) / 4;       // 4 waves (rows) per 256-thread block
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#define DISPATCH_PER(group, CALL)                       \
    do {                                                \
        int per_ = ((group) + 63) / 64;                 \
        if (per_ <= 1) { CALL(1); }                     \
        else if (per_ <= 2) { CALL(2); }                \
        else if (per_ <= 4) { CALL(4); }                \
        else if (per_ <= 8) { CALL(8); }                \
        else if (per_ <= 16) { CALL(16); }              \
        else if (per_ <= 40) { CALL(40); }              \
        else if (per_ <= 128) { CALL(128); }            \
        else return ASR_EUNSUPPORTED;                   \
    } while (0)

}  // namespace

extern "C" int asr_log_softmax_fwd_f32(const float *x, int64_t rows, int group,
                                       float *y, void *stream) {
    if (rows < 0 || group <= 0) return ASR_EINVAL;
    if (rows == 0) return ASR_OK;
    if (!x || !y) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(P) hipLaunchKernelGGL(log_softmax_fwd_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, x, y, rows, group)
    DISPATCH_PER(group, CALL);
#undef CALL
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_log_softmax_bwd_f32(const float *y, const float *dy,
                                       int64_t rows, int group, float *dx,
                                       void *stream) {
    if (rows < 0 || group <= 0) return ASR_EINVAL;
    if (rows == 0) return ASR_OK;
    if (!y || !dy || !dx) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(P) hipLaunchKernelGGL(log_softmax_bwd_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, y, dy, dx, rows, group)
    DISPATCH_PER(group, CALL);
#undef CALL
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_sub_rowmax_f32(const float *x, int T, int B, int C,
                                  const int32_t *lens, float *y, float *row_max,
                                  float *max_sum, void *stream) {
    if (T < 0 || B < 0 || C <= 0) return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!lens || !max_sum) return ASR_EINVAL;
    if (T > 0 && (!x || !y || !row_max)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)T * B;
    if (T > 0) {
#define CALL(P) hipLaunchKernelGGL(sub_rowmax_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, x, y, row_max, T, B, C)
        DISPATCH_PER(C, CALL);
#undef CALL
    }
    hipLaunchKernelGGL(max_sum_kernel, dim3(B), dim3(64), 0, s, row_max, lens, max_sum, T, B);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_argmax_rows_f32(const float *x, int64_t rows, int C,
                                   int32_t *out_idx, void *stream) {
    if (rows < 0 || C <= 0) return ASR_EINVAL;
    if (rows == 0) return ASR_OK;
    if (!x || !out_idx) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(P) hipLaunchKernelGGL(argmax_rows_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, x, out_idx, rows, C)
    DISPATCH_PER(C, CALL);
#undef CALL
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}


// out[e] = sum_g in[g][e] for G stacked fp32 images of n elements (n % 4 == 0): the partial
// products of the chunked weight-gradient GEMMs (att_speech/modules/encoders/native_lstm.py).
// One float4 per thread per image, all G loads of a thread independent: HBM-bound.
namespace {
__global__ __launch_bounds__(256) void sum_leading_kernel(const float4 *in, int G, int64_t n4, float4 *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = in[i];
    for (int g = 1; g < G; ++g) {
        const float4 v = in[(int64_t)g * n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
}
}  // namespace

extern "C" int asr_sum_leading_f32(const float *in, int G, int64_t n, float *out, void *stream) {
    if (!in || !out || G <= 0 || n <= 0 || (n & 3)) return ASR_EINVAL;
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4 *)in, G, n4, (float4 *)out);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}


// -----------------------------------------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): the operands of the split-bf16 class
// projection (three bf16 MFMA products hi hi + hi lo + lo hi stand for one fp32 product,
// relative error 2^-16 per term).  One pass: 4 bytes in, 2 + 2 out per element.
namespace {
__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) {
    return __builtin_bit_cast(float, (unsigned)h << 16);
}
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *x, int64_t rows, int64_t cols, int64_t ldx,
                                                         unsigned short *hi, int64_t ldhi,
                                                         unsigned short *lo, int64_t ldlo, int vec) {
    const int64_t q = (cols + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * q) return;
    const int64_t r = i / q, c = (i - r * q) * 4;
    const float *xp = x + r * ldx + c;
    unsigned short *hp = hi + r * ldhi + c, *lp = lo + r * ldlo + c;
    if (vec && c + 3 < cols) {
        const float4 v = *reinterpret_cast<const float4 *>(xp);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = bf16_rne(f[k]);
            l[k] = bf16_rne(f[k] - bf16_f32(h[k]));
        }
        *reinterpret_cast<uint2 *>(hp) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        *reinterpret_cast<uint2 *>(lp) = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
    } else {
        for (int k = 0; k < 4 && c + k < cols; ++k) {
            const unsigned short h = bf16_rne(xp[k]);
            hp[k] = h;
            lp[k] = bf16_rne(xp[k] - bf16_f32(h));
        }
    }
}
// dense x (ldx == cols) whose rows are NOT 16-byte aligned (2401 classes): 16-byte loads over
// the flat tensor, the four elements of a load find their own (row, column)
__global__ __launch_bounds__(256) void split_bf16_flat_kernel(const float *x, int64_t n, int64_t cols,
                                                              unsigned short *hi, int64_t ldhi,
                                                              unsigned short *lo, int64_t ldlo) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        for (int k = 0; i + k < n; ++k) f[k] = x[i + k];
    }
    int64_t r = i / cols, c = i - r * cols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k < n) {
            const unsigned short h = bf16_rne(f[k]);
            hi[r * ldhi + c] = h;
            lo[r * ldlo + c] = bf16_rne(f[k] - bf16_f32(h));
        }
        if (++c == cols) { c = 0; ++r; }
    }
}
}  // namespace

extern "C" int asr_split_bf16_f32(const float *x, int64_t rows, int64_t cols, int64_t ldx,
                                  void *hi_bf16, int64_t ldhi, void *lo_bf16, int64_t ldlo, void *stream) {
    if (rows < 0 || cols < 0 || ldx < cols || ldhi < cols || ldlo < cols) return ASR_EINVAL;
    if (rows == 0 || cols == 0) return ASR_OK;
    if (!x || !hi_bf16 || !lo_bf16) return ASR_EINVAL;
    const int64_t items = rows * ((cols + 3) >> 2);
    if (items > (int64_t)0x7fffffff * 256) return ASR_EUNSUPPORTED;
    // 16-byte loads / 8-byte stores when every row starts aligned
    const int vec = ((ldx | ldhi | ldlo) & 3) == 0 && ((uintptr_t)x & 15) == 0 &&
                    (((uintptr_t)hi_bf16 | (uintptr_t)lo_bf16) & 7) == 0;
    if (!vec && ldx == cols && rows > 1 && ((uintptr_t)x & 15) == 0) {
        const int64_t n = rows * cols, it = (n + 3) >> 2;
        hipLaunchKernelGGL(split_bf16_flat_kernel, dim3((unsigned)((it + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, n, cols, (unsigned short *)hi_bf16, ldhi,
                           (unsigned short *)lo_bf16, ldlo);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, rows, cols, ldx, (unsigned short *)hi_bf16, ldhi, (unsigned short *)lo_bf16, ldlo, vec);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}


// -----------------------------------------------------------------------------------------
// Normalise + stabilise in one pass (FSTDecoder with normalize_by_dim = 0 and the row-max
// stabilisation of advanced_decoder.py:479-484): log_softmax(x) - max_c log_softmax(x) is
// x - max_c x — the normaliser cancels — so the shifted acts are written straight from the
// logits and the normaliser only enters the per-row constant
//     nls[r] = max_c log_softmax(x[r]) = -log sum_c exp(x[r,c] - max_c x[r]),
// whose masked sum over frames is the 'denominator' the reference subtracts.  The backward
// pass needs softmax(x) = exp(shifted + nls): it is recomputed, no normalised copy is kept.
// One read + one write of [rows, C] forward instead of two of each.
// -----------------------------------------------------------------------------------------
namespace {
using namespace asr;

template <int PER>
__global__ void lsm_shift_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                     float *__restrict__ nls, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *xr = x + r * C;
        float v[PER];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            v[i] = c < C ? xr[c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float s = 0.f;
        float *yr = y + r * C;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            const float d = v[i] - m;
            s += __expf(d);
            if (c < C) yr[c] = d;
        }
        s = wave_sum(s);
        if (lane == 0) nls[r] = -__logf(s);
    }
}

template <int PER>
__global__ void lsm_shift_bwd_kernel(const float *__restrict__ y, const float *__restrict__ nls,
                                     const float *__restrict__ dy, float *__restrict__ dx,
                                     int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *yr = y + r * C, *gr = dy + r * C;
        const float off = nls[r];
        float vy[PER], vg[PER];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            vy[i] = c < C ? yr[c] : -INFINITY;
            vg[i] = c < C ? gr[c] : 0.f;
            s += vg[i];
        }
        s = wave_sum(s);
        float *dr = dx + r * C;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            if (c < C) dr[c] = vg[i] - __expf(vy[i] + off) * s;
        }
    }
}

// The same gradient handed to the split-bf16 class projection: dx = hi + lo as two bf16
// tensors [rows, ld] (columns >= C zero) instead of fp32 — the fp32 tensor (1.6 GB at 2401
// classes) is never written nor read back by a separate split pass — and the column sums of dx
// (the projection's bias gradient) as one partial row per workgroup, summed by the caller.
template <int PER>
__global__ __launch_bounds__(256) void lsm_shift_bwd_split_kernel(
    const float *__restrict__ y, const float *__restrict__ nls, const float *__restrict__ dy,
    unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, float *__restrict__ colsum,
    int64_t rows, int C, int ld) {
    __shared__ float red[4][PER * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float cs[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) cs[i] = 0.f;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *yr = y + r * C, *gr = dy + r * C;
        const float off = nls[r];
        float vy[PER], vg[PER];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            vy[i] = c < C ? yr[c] : -INFINITY;
            vg[i] = c < C ? gr[c] : 0.f;
            s += vg[i];
        }
        s = wave_sum(s);
        unsigned short *hr = hi + r * ld, *lr = lo + r * ld;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 64 + lane;
            if (c < ld) {
                const float d = c < C ? vg[i] - __expf(vy[i] + off) * s : 0.f;
                const unsigned short h = bf16_rne(d);
                hr[c] = h;
                lr[c] = bf16_rne(d - bf16_f32(h));
                cs[i] += d;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) red[wv][i * 64 + lane] = cs[i];
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)
        colsum[(int64_t)blockIdx.x * ld + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
}  // namespace

static int split_blocks(int64_t rows) {
    const int64_t b = (rows + 3) / 4;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
extern "C" int asr_log_softmax_shift_bwd_split_blocks(int64_t rows) { return rows > 0 ? split_blocks(rows) : 0; }

extern "C" int asr_log_softmax_shift_bwd_split_bf16(const float *y, const float *nls, const float *dy,
                                                    int64_t rows, int C, void *hi_bf16, void *lo_bf16, int ld,
                                                    float *colsum_partial, void *stream) {
    if (rows < 0 || C <= 0 || ld < C) return ASR_EINVAL;
    if (rows == 0) return ASR_OK;
    if (!y || !nls || !dy || !hi_bf16 || !lo_bf16 || !colsum_partial) return ASR_EINVAL;
    if (ld > 64 * 40) return ASR_EUNSUPPORTED;          // (the kernel's per-lane column count)
    hipStream_t s = (hipStream_t)stream;
    // PER covers the padded row (ld columns)
#define CALL(P) hipLaunchKernelGGL(lsm_shift_bwd_split_kernel<P>, dim3(split_blocks(rows)), dim3(256), 0, s, y, nls, dy, \
                                   (unsigned short *)hi_bf16, (unsigned short *)lo_bf16, colsum_partial, rows, C, ld)
    DISPATCH_PER(ld, CALL);
#undef CALL
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_log_softmax_shift_fwd_f32(const float *x, int T, int B, int C,
                                             const int32_t *lens, float *y, float *nls,
                                             float *nls_sum, void *stream) {
    if (T < 0 || B < 0 || C <= 0) return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!lens || !nls_sum) return ASR_EINVAL;
    if (T > 0 && (!x || !y || !nls)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)T * B;
    if (T > 0) {
#define CALL(P) hipLaunchKernelGGL(lsm_shift_fwd_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, x, y, nls, rows, C)
        DISPATCH_PER(C, CALL);
#undef CALL
    }
    hipLaunchKernelGGL(max_sum_kernel, dim3(B), dim3(64), 0, s, nls, lens, nls_sum, T, B);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_log_softmax_shift_bwd_f32(const float *y, const float *nls, const float *dy,
                                             int64_t rows, int C, float *dx, void *stream) {
    if (rows < 0 || C <= 0) return ASR_EINVAL;
    if (rows == 0) return ASR_OK;
    if (!y || !nls || !dy || !dx) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(P) hipLaunchKernelGGL(lsm_shift_bwd_kernel<P>, dim3(grid_for(rows)), dim3(256), 0, s, y, nls, dy, dx, rows, C)
    DISPATCH_PER(C, CALL);
#undef CALL
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
