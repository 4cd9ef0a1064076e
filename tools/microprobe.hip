// Development probe: shader clock, dependent-op latencies, LDS round trip and
// s_barrier cost on gfx950.  hipcc --offload-arch=gfx950 -O3 -o microprobe microprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 4096

__global__ void probe(float *out, unsigned long long *tm, int mode) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    float v = out[tid] + 1.0f;
    lds[tid] = v;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {            // dependent v_add chain
#pragma unroll 16
        for (int i = 0; i < N_IT; ++i) v = v + 1.25f;
    } else if (mode == 1) {     // dependent exp2 chain
#pragma unroll 16
        for (int i = 0; i < N_IT; ++i) v = __builtin_amdgcn_exp2f(v) - 0.5f;
    } else if (mode == 2) {     // LDS write -> read own (no barrier)
#pragma unroll 8
        for (int i = 0; i < N_IT; ++i) { lds[tid] = v; asm volatile("s_waitcnt lgkmcnt(0)"); v = lds[tid ^ 1] + 1.0f; }
    } else if (mode == 3) {     // LDS write -> barrier -> read neighbour
#pragma unroll 8
        for (int i = 0; i < N_IT; ++i) { lds[tid] = v; __syncthreads(); v = lds[(tid + 1) % blockDim.x] + 1.0f; __syncthreads(); }
    } else if (mode == 4) {     // one barrier per iter with double buffer (like the scan)
#pragma unroll 8
        for (int i = 0; i < N_IT; ++i) { lds[(i & 1) * 1024 + tid] = v; __syncthreads(); v = lds[(i & 1) * 1024 + ((tid + 1) % blockDim.x)] + 1.0f; }
    } else if (mode == 5) {     // full LSE3 step with double buffer + barrier
        const int NT = blockDim.x;
#pragma unroll 8
        for (int i = 0; i < N_IT; ++i) {
            const float *s = lds + (i & 1) * 1024;
            float a = s[tid], b = s[(tid + NT - 1) % NT], c = s[(tid + NT - 2) % NT];
            float m = fmaxf(a, fmaxf(b, c));
            float su = __builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m);
            v = m + __builtin_amdgcn_logf(su) - 1.5f;
            lds[((i & 1) ^ 1) * 1024 + tid] = v;
            __syncthreads();
        }
    } else if (mode == 6) {     // taken-branch cost: loop of tiny blocks with scalar branches
        int k = 0;
        for (int i = 0; i < N_IT; ++i) { asm volatile("s_nop 0" ::: "memory"); k += i; }
        v += k;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = v;
    if (tid == 0) { tm[blockIdx.x * 2] = t1 - t0; tm[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    float *out; unsigned long long *tm;
    hipMalloc(&out, 4096 * 4); hipMemset(out, 0, 4096 * 4);
    hipMalloc(&tm, 1024 * 16);
    const char *names[] = {"dep v_add", "dep exp2+sub", "lds wr->rd (no barrier)", "lds wr,bar,rd,bar", "lds dbuf 1 barrier", "LSE3 step dbuf+barrier", "loop+s_nop"};
    for (int nt : {64, 256, 512}) {
        for (int blocks : {1, 512}) {
            for (int mode = 0; mode < 7; ++mode) {
                for (int rep = 0; rep < 3; ++rep)
                    hipLaunchKernelGGL(probe, dim3(blocks), dim3(nt), 8192 + 64, 0, out, tm, mode);
                hipDeviceSynchronize();
                unsigned long long h[2];
                hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
                double cyc = (double)h[0] / N_IT, ns = (double)h[1] * 10.0 / N_IT;
                printf("nt=%4d blocks=%4d %-26s %8.1f clk/iter %8.1f ns/iter  clock %.2f GHz\n", nt, blocks, names[mode], cyc, ns, cyc / ns);
            }
        }
    }
    return 0;
}
