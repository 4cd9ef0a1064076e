// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASR_WAVE 64

namespace asr {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, ASR_WAVE));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ASR_WAVE);
    return v;
}

// Block-wide reductions through a small LDS scratch (>= 32 floats).
// Every thread of the block must call; result is returned to every thread.
__device__ __forceinline__ float block_max(float v, float *scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

__device__ __forceinline__ float block_sum(float v, float *scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r += scratch[i];
    return r;
}

// online log-sum-exp accumulator: (m, s) with value m + log(s)
struct Lse {
    float m, s;
    __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; }
    __device__ __forceinline__ void add(float v) {
        if (v > m) {
            s = s * __expf(m - v) + 1.f;
            m = v;
        } else {
            s += __expf(v - m);
        }
    }
    __device__ __forceinline__ float value() const { return m + __logf(s); }
};

}  // namespace asr
