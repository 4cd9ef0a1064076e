// Bidirectional LSTM recurrence (no bias) for gfx950 — the recurrent part of
// the reference's BatchRNN (att_speech/modules/encoders/encoder_utils.py:55-124:
// nn.LSTM(bidirectional=True, bias=False) on a packed batch).
//
// The input projection x_t·W_ihᵀ of all frames and both directions is one
// dense GEMM done by the caller; this file owns the sequential part:
//   gates_t = gx_t + h_{t-1}·W_hhᵀ ;  i,f,o = σ(.), g = tanh(.)
//   c_t = f·c_{t-1} + i·g ;  h_t = o·tanh(c_t)
// One launch per time step covers BOTH directions (forward frame s, reverse
// frame T-1-s) so the grid has 2·(B/32)·(H/32) workgroups.  A workgroup owns a
// [32 batch x 32 hidden] tile: wave g computes the pre-activation of gate g
// with v_mfma_f32_32x32x16_bf16 (K = H), operands are read from L2 straight in
// MFMA fragment order (16 B per lane, k-contiguous; W_hh is 0.8 MB per
// direction and stays L2-resident across the steps), then the four gate tiles
// meet in LDS for the pointwise cell update in fp32.
//
// Packed-sequence semantics with a padded batch: utterance b is active at
// frame t iff t < lens[b]; inactive frames keep (h, c) and emit zeros, so the
// reverse direction starts from the zero state at each utterance's own last
// frame exactly like pack_padded_sequence does.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.f * fabsf(x));
    const float t = (1.f - e) / (1.f + e);
    return x < 0.f ? -t : t;
}

struct LstmFwdParams {
    const float *gx;        // [T,B,2,4H] x·W_ihᵀ, gate order i,f,g,o
    const __bf16 *whh;      // [2,4H,H]
    const int32_t *lens;    // [B]
    int T, B, H;
    __bf16 *hbuf;           // [2 pingpong][2 dir][B][H]
    float *cbuf;            // [2 dir][B][H]
    float *y;               // [T,B,2,H] per-direction outputs (zeros when inactive)
    float *gates;           // [T,2,B,4,H] post-activation gates (saved for backward)
    float *csave;           // [T,2,B,H] cell state after the step
    int step;
};

// grid: x = hidden tile (H/32), y = batch tile (ceil(B/32)), z = direction
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmFwdParams p) {
    __shared__ float g_lds[4][32][33];
    const int H = p.H, B = p.B;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32, dir = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = dir == 0 ? p.step : p.T - 1 - p.step;
    const __bf16 *hprev = p.hbuf + ((size_t)(p.step & 1) * 2 + dir) * B * H;
    __bf16 *hnext = p.hbuf + ((size_t)((p.step + 1) & 1) * 2 + dir) * B * H;

    // ---- gate pre-activation tile: [32 batch] x [32 hidden of gate `wave`]
    {
        const int r = lane & 31, kh = (lane >> 5) * 8;
        int brow = b0 + r;
        if (brow >= B) brow = B - 1;                    // clamp (masked at the store)
        const __bf16 *ap = hprev + (size_t)brow * H + kh;
        const __bf16 *bp = p.whh + ((size_t)dir * 4 * H + (size_t)wave * H + j0 + r) * H + kh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int k = 0; k < H; k += 16) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(ap + k);
            const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(bp + k);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        const int col = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            g_lds[wave][row][col] = acc[i];
        }
    }
    __syncthreads();

    // ---- pointwise cell update: 1024 (b, j) elements over 256 threads
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int row = idx >> 5, col = idx & 31;
        const int b = b0 + row, j = j0 + col;
        if (b < B) {
        const bool active = t < p.lens[b];
        const size_t sidx = ((size_t)dir * B + b) * H + j;
        float *yo = p.y + (((size_t)t * B + b) * 2 + dir) * H + j;
        const size_t gsave = ((((size_t)t * 2 + dir) * B + b) * 4) * H + j;
        const size_t csv = (((size_t)t * 2 + dir) * B + b) * H + j;
        if (active) {
            const float *gxp = p.gx + (((size_t)t * B + b) * 2 + dir) * 4 * H + j;
            const float gi = sigmoidf_(g_lds[0][row][col] + gxp[0]);
            const float gf = sigmoidf_(g_lds[1][row][col] + gxp[H]);
            const float gg = tanhf_(g_lds[2][row][col] + gxp[2 * H]);
            const float go = sigmoidf_(g_lds[3][row][col] + gxp[3 * H]);
            const float c = gf * p.cbuf[sidx] + gi * gg;
            const float h = go * tanhf_(c);
            p.cbuf[sidx] = c;
            hnext[(size_t)b * H + j] = (__bf16)h;
            *yo = h;
            p.gates[gsave] = gi;
            p.gates[gsave + H] = gf;
            p.gates[gsave + 2 * H] = gg;
            p.gates[gsave + 3 * H] = go;
            p.csave[csv] = c;
        } else {
            hnext[(size_t)b * H + j] = hprev[(size_t)b * H + j];
            *yo = 0.f;
            p.csave[csv] = p.cbuf[sidx];
        }
        }
    }
}

struct LstmBwdParams {
    const float *dy;        // [T,B,2,H] gradient w.r.t. the per-direction outputs
    const __bf16 *whhT;     // [2,H,4H]  (W_hh transposed: k-contiguous for dgates·W_hh)
    const int32_t *lens;
    int T, B, H;
    const float *gates;     // [T,2,B,4,H]
    const float *csave;     // [T,2,B,H]
    __bf16 *dgbuf;          // [2 pingpong][2 dir][B][4H] dgates of the previous step
    float *dcbuf;           // [2 dir][B][H] carried dL/dc
    float *dgates;          // [T,B,2,4H] pre-activation gate gradients (output)
    int step;
};

__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmBwdParams p) {
    __shared__ float part[4][32][33];
    const int H = p.H, B = p.B, H4 = 4 * p.H;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32, dir = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the backward scan visits frames in the opposite order of the forward one
    const int t = dir == 0 ? p.T - 1 - p.step : p.step;
    const __bf16 *dgprev = p.dgbuf + ((size_t)(p.step & 1) * 2 + dir) * B * H4;
    __bf16 *dgnext = p.dgbuf + ((size_t)((p.step + 1) & 1) * 2 + dir) * B * H4;

    // ---- dh_rec[b][j] = sum_k dgates_prev[b][k] * W_hh[k][j]; K = 4H split
    // over the four waves (wave w takes the columns of gate w)
    {
        const int r = lane & 31, kh = (lane >> 5) * 8;
        int brow = b0 + r;
        if (brow >= B) brow = B - 1;
        const __bf16 *ap = dgprev + (size_t)brow * H4 + (size_t)wave * H + kh;
        const __bf16 *bp = p.whhT + ((size_t)dir * H + j0 + r) * H4 + (size_t)wave * H + kh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int k = 0; k < H; k += 16) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(ap + k);
            const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(bp + k);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        const int col = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            part[wave][row][col] = acc[i];
        }
    }
    __syncthreads();

#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int row = idx >> 5, col = idx & 31;
        const int b = b0 + row, j = j0 + col;
        if (b < B) {
        const int len = p.lens[b];
        const bool active = t < len;
        const size_t sidx = ((size_t)dir * B + b) * H + j;
        float *dgo = p.dgates + (((size_t)t * B + b) * 2 + dir) * H4 + j;
        __bf16 *dgn = dgnext + (size_t)b * H4 + j;
        if (active) {
            const float dh = p.dy[(((size_t)t * B + b) * 2 + dir) * H + j] +
                             (part[0][row][col] + part[1][row][col]) +
                             (part[2][row][col] + part[3][row][col]);
            const size_t gsave = ((((size_t)t * 2 + dir) * B + b) * 4) * H + j;
            const float gi = p.gates[gsave], gf = p.gates[gsave + H];
            const float gg = p.gates[gsave + 2 * H], go = p.gates[gsave + 3 * H];
            const float c = p.csave[(((size_t)t * 2 + dir) * B + b) * H + j];
            // cell state the step started from
            const int tp = dir == 0 ? t - 1 : t + 1;
            float cprev = 0.f;
            if (tp >= 0 && tp < len)
                cprev = p.csave[(((size_t)tp * 2 + dir) * B + b) * H + j];
            const float tc = tanhf_(c);
            const float dc = dh * go * (1.f - tc * tc) + p.dcbuf[sidx];
            const float d_o = dh * tc * go * (1.f - go);
            const float d_i = dc * gg * gi * (1.f - gi);
            const float d_f = dc * cprev * gf * (1.f - gf);
            const float d_g = dc * gi * (1.f - gg * gg);
            p.dcbuf[sidx] = dc * gf;
            dgo[0] = d_i; dgo[H] = d_f; dgo[2 * H] = d_g; dgo[3 * H] = d_o;
            dgn[0] = (__bf16)d_i; dgn[H] = (__bf16)d_f;
            dgn[2 * H] = (__bf16)d_g; dgn[3 * H] = (__bf16)d_o;
        } else {
            // no gradient reaches a padding frame; the carried state gradient
            // restarts from zero (forward: beyond the end; reverse: before the start)
            p.dcbuf[sidx] = 0.f;
            dgo[0] = 0.f; dgo[H] = 0.f; dgo[2 * H] = 0.f; dgo[3 * H] = 0.f;
            dgn[0] = (__bf16)0.f; dgn[H] = (__bf16)0.f;
            dgn[2 * H] = (__bf16)0.f; dgn[3 * H] = (__bf16)0.f;
        }
        }
    }
}

__global__ void zero_bytes_kernel(uint32_t *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0u;
}

inline void zero_async(void *p, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 4;
    if (!n) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t *)p, n);
}

}  // namespace

extern "C" int64_t asr_lstm_workspace_bytes(int B, int H) {
    if (B < 0 || H < 0) return -1;
    // forward: hbuf bf16 [2][2][B][H] + cbuf f32 [2][B][H]
    // backward: dgbuf bf16 [2][2][B][4H] + dcbuf f32 [2][B][H]   (the larger one)
    return (int64_t)2 * 2 * B * 4 * H * 2 + (int64_t)2 * B * H * 4 + 256;
}

extern "C" int asr_lstm_bidir_fwd_bf16(const float *gx, const void *whh_bf16,
                                       const int32_t *lens, int T, int B, int H,
                                       float *y, float *gates, float *csave,
                                       void *workspace, int64_t workspace_bytes,
                                       void *stream) {
    if (T < 0 || B <= 0 || H <= 0 || (H % 32) != 0) return ASR_EINVAL;
    if (T == 0) return ASR_OK;
    if (!gx || !whh_bf16 || !lens || !y || !gates || !csave || !workspace) return ASR_EINVAL;
    if (workspace_bytes < asr_lstm_workspace_bytes(B, H)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    LstmFwdParams p;
    p.gx = gx; p.whh = (const __bf16 *)whh_bf16; p.lens = lens;
    p.T = T; p.B = B; p.H = H;
    p.hbuf = (__bf16 *)workspace;
    p.cbuf = (float *)((char *)workspace + (size_t)2 * 2 * B * H * 2);
    p.y = y; p.gates = gates; p.csave = csave;
    zero_async(workspace, (size_t)2 * 2 * B * H * 2 + (size_t)2 * B * H * 4, s);
    const dim3 grid(H / 32, (B + 31) / 32, 2);
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_lstm_bidir_bwd_bf16(const float *dy, const void *whhT_bf16,
                                       const int32_t *lens, int T, int B, int H,
                                       const float *gates, const float *csave,
                                       float *dgates,
                                       void *workspace, int64_t workspace_bytes,
                                       void *stream) {
    if (T < 0 || B <= 0 || H <= 0 || (H % 32) != 0) return ASR_EINVAL;
    if (T == 0) return ASR_OK;
    if (!dy || !whhT_bf16 || !lens || !gates || !csave || !dgates || !workspace)
        return ASR_EINVAL;
    if (workspace_bytes < asr_lstm_workspace_bytes(B, H)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    LstmBwdParams p;
    p.dy = dy; p.whhT = (const __bf16 *)whhT_bf16; p.lens = lens;
    p.T = T; p.B = B; p.H = H;
    p.gates = gates; p.csave = csave;
    p.dgbuf = (__bf16 *)workspace;
    p.dcbuf = (float *)((char *)workspace + (size_t)2 * 2 * B * 4 * H * 2);
    p.dgates = dgates;
    zero_async(workspace, (size_t)2 * 2 * B * 4 * H * 2 + (size_t)2 * B * H * 4, s);
    const dim3 grid(H / 32, (B + 31) / 32, 2);
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
