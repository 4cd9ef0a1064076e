// Bidirectional LSTM recurrence (no bias) for gfx950 — the recurrent part of
// the reference's BatchRNN (att_speech/modules/encoders/encoder_utils.py:55-124:
// nn.LSTM(bidirectional=True, bias=False) on a packed batch).
//
// The input projection x_t·W_ihᵀ of all frames and both directions is one
// dense GEMM done by the caller; this file owns the sequential part:
//   gates_t = gx_t + h_{t-1}·W_hhᵀ ;  i,f,o = σ(.), g = tanh(.)
//   c_t = f·c_{t-1} + i·g ;  h_t = o·tanh(c_t)
// Two implementations of the same arithmetic (bit-identical outputs, see
// tests/test_lstm_gpu.py):
//   * PERSISTENT (default): one launch walks all T steps; a 512-thread workgroup
//     keeps its W_hh fragments and the cell state in registers for the whole
//     sequence and teams of H/64 workgroups hand h_t / dgates_t over through L2
//     once per step (sc1 write-through + agent-scope counters) — further down;
//   * PER-STEP (ASR_LSTM_PERSIST=0, and the fallback for shapes the persistent
//     kernels do not take): one launch per time step covering BOTH directions
//     (forward frame s, reverse frame T-1-s), grid 2·(B/32)·(H/32); a workgroup
//     owns a [32 batch x 32 hidden] tile, wave g computes the pre-activation of
//     gate g with v_mfma_f32_32x32x16_bf16 (K = H), operands read from L2 straight
//     in MFMA fragment order (W_hh is 0.8 MB per direction and stays L2-resident
//     across the steps), the four gate tiles meet in LDS for the fp32 cell update.
//
// MFMA operand layout: both GEMM operands live in memory in FRAGMENT-MAJOR
// order — for a [32 rows x K] tile, k-step ks is the 1 KiB block
// [64 lanes][8 bf16] with lane l = (row & 31) + 32*((k>>3)&1) — so a wave's
// fragment load is one fully coalesced 1 KiB access (a row-major tile makes
// every load touch 32 cache lines and the address coalescer, not the MFMA,
// sets the pace: measured 9k cycles just to ISSUE the 40 loads).  The h /
// dgates ping-pong buffers are written in that order by the pointwise phase;
// W_hh is re-packed once per call by lstm_pack_kernel.
//
// Packed-sequence semantics with a padded batch: utterance b is active at
// frame t iff t < lens[b]; inactive frames keep (h, c) and emit zeros, so the
// reverse direction starts from the zero state at each utterance's own last
// frame exactly like pack_padded_sequence does.
#include "common.h"
#include "../../include/asr_amd.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// The four post-activation gates of one (frame, direction, utterance, hidden unit) are
// saved for the backward pass as ONE 8-byte record of four bf16 (i,f,g,o): a single
// coalesced 8 B/lane store / load per element instead of four fp32 ones with stride H
// (the saved gates were the recurrence's largest stream: 1.75 GB per layer at B=512).
__device__ __forceinline__ u32x2 pack_gates(const float g[4]) {
    const unsigned s0 = __builtin_bit_cast(unsigned short, (__bf16)g[0]);
    const unsigned s1 = __builtin_bit_cast(unsigned short, (__bf16)g[1]);
    const unsigned s2 = __builtin_bit_cast(unsigned short, (__bf16)g[2]);
    const unsigned s3 = __builtin_bit_cast(unsigned short, (__bf16)g[3]);
    u32x2 v;
    v[0] = s0 | (s1 << 16);
    v[1] = s2 | (s3 << 16);
    return v;
}
__device__ __forceinline__ void unpack_gates(u32x2 v, float g[4]) {
    const unsigned lo = v[0], hi = v[1];
    g[0] = __builtin_bit_cast(float, lo << 16);
    g[1] = __builtin_bit_cast(float, lo & 0xFFFF0000u);
    g[2] = __builtin_bit_cast(float, hi << 16);
    g[3] = __builtin_bit_cast(float, hi & 0xFFFF0000u);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;

// v_exp_f32 / v_rcp_f32 forms (about 1 ulp each): four instructions per
// sigmoid instead of a full-precision division sequence
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}

// The cell update and its derivative, shared by the per-step and the persistent
// kernels.  Contraction is off so both kernels round identically (the tests
// compare them bit for bit to catch a stale hand-off).
__device__ __forceinline__ void lstm_cell_fwd(const float pre[4], float cprev, float gate[4],
                                              float &c, float &h) {
#pragma clang fp contract(off)
    gate[0] = sigmoidf_(pre[0]);
    gate[1] = sigmoidf_(pre[1]);
    gate[2] = tanhf_(pre[2]);
    gate[3] = sigmoidf_(pre[3]);
    c = gate[1] * cprev + gate[0] * gate[2];
    h = gate[3] * tanhf_(c);
}

// Two elements at a time: the multiplies and adds around the transcendentals as packed fp32
// operations (v_pk_mul_f32 / v_pk_add_f32), half the instructions; the same IEEE operations in
// the same order per element, so the results are bit for bit those of lstm_cell_fwd.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 sigmoid2_(f32x2 x) {
#pragma clang fp contract(off)
    const f32x2 t = x * -1.4426950408889634f;
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.f;
    return f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
__device__ __forceinline__ f32x2 tanh2_(f32x2 x) {
#pragma clang fp contract(off)
    const f32x2 t = x * -2.8853900817779268f;
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return r * 2.f - 1.f;
}
__device__ __forceinline__ void lstm_cell_fwd2(const f32x2 pre[4], f32x2 cprev, f32x2 gate[4],
                                               f32x2 &c, f32x2 &h) {
#pragma clang fp contract(off)
    gate[0] = sigmoid2_(pre[0]);
    gate[1] = sigmoid2_(pre[1]);
    gate[2] = tanh2_(pre[2]);
    gate[3] = sigmoid2_(pre[3]);
    c = gate[1] * cprev + gate[0] * gate[2];
    h = gate[3] * tanh2_(c);
}

// dh: gradient reaching h_t; dcin: carried dL/dc_t from the later step; cs: c_t;
// cp: c_{t-1}.  d[] = pre-activation gate gradients (i, f, g, o).
__device__ __forceinline__ void lstm_cell_bwd(const float g[4], float cs, float cp, float dh,
                                              float dcin, float d[4], float &dcout) {
#pragma clang fp contract(off)
    const float gi = g[0], gf = g[1], gg = g[2], go = g[3];
    const float tc = tanhf_(cs);
    const float dc = dh * go * (1.f - tc * tc) + dcin;
    d[3] = dh * tc * go * (1.f - go);
    d[0] = dc * gg * gi * (1.f - gi);
    d[1] = dc * cp * gf * (1.f - gf);
    d[2] = dc * gi * (1.f - gg * gg);
    dcout = dc * gf;
}

// x·W_ih term: fp32, or the bf16 the input-projection GEMM emits (halves its 1.75 GB
// output at B=512 and this kernel's reads; fp32 accumulation either way)
template <int GXB>
__device__ __forceinline__ float load_gx(const void *gx, size_t i) {
    if constexpr (GXB) return (float)((const __bf16 *)gx)[i];
    else return ((const float *)gx)[i];
}

struct LstmFwdParams {
    const void *gx;         // [T,B,2,4H] x·W_ihᵀ, gate order i,f,g,o; float or (gx_bf16) __bf16
    int gx_bf16;
    const __bf16 *x;        // fused input projection (persistent kernel, F == H): x [T,B,H] bf16 row-major
    const __bf16 *wih;      // ... and the fragment-major pack of W_ih [2*4 (dir,gate)][H rows][F cols]
    const __bf16 *whh;      // fragment-major pack of [2*4 (dir,gate)][H rows][H cols]
    const int32_t *lens;    // [B]
    int T, B, H;
    __bf16 *hbuf;           // [2 pingpong][2 dir] fragment-major [Bp x H], Bp = B padded to 32
    float *cbuf;            // [2 dir][B][H]
    float *y;               // [T,B,2,H] per-direction outputs (zeros when inactive)
    __bf16 *ybf;            // [2,T+2,B,H] bf16 copy, frame t at index t+1 (zero frames at both ends)
    u32x2 *gates;           // [T,2,B,H] records of 4 bf16 post-activation gates (saved for backward)
    float *csave;           // [T,2,B,H] cell state after the step
    int step;
    // persistent kernel: steps [s_begin, s_end) of the T (a launch that starts past step 0
    // picks its state up from hbuf / csave, the team counters keep counting).  xsum [T,B,H]
    // bf16 != null: every step also writes bf16(h_own) + the OTHER direction's bf16 output of
    // its frame — valid when that direction wrote the frame in an EARLIER launch.
    int s_begin, s_end;
    __bf16 *xsum;
};

// K-loop of one wave: acc += A[32 x 16*KS] * B[16*KS x 32] with both operands
// read from global memory (L2) in MFMA fragment order.  With KS known at
// compile time the loop is fully unrolled and every fragment load is issued
// before the first MFMA, so the L2 latency is paid once, not per k-step.
// element (row b, column k) of a [rows x 16*KS] operand in fragment-major order
__device__ __forceinline__ size_t frag_off(int b, int k, int KS) {
    return ((((size_t)(b >> 5) * KS + (k >> 4)) * 64 + ((b & 31) + 32 * ((k >> 3) & 1))) << 3) + (k & 7);
}

// out = fragment-major copy of `rows` x `cols` (cols % 16 == 0, rows % 32 == 0)
// row-major bf16 matrices, `nmat` of them; transpose != 0 reads in[c][r].
__global__ void lstm_pack_kernel(const __bf16 *in, __bf16 *out, int nmat, int rows,
                                 int cols, int transpose) {
    const size_t per = (size_t)rows * cols;
    const size_t total = per * nmat;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / per);
        const size_t rem = i % per;
        const int r = (int)(rem / cols), c = (int)(rem % cols);
        const __bf16 v = transpose ? in[(size_t)m * per + (size_t)c * rows + r]
                                   : in[(size_t)m * per + rem];
        out[(size_t)m * per + frag_off(r, c, cols / 16)] = v;
    }
}

// grid: x = hidden tile (H/32), y = batch tile group (ceil(Bp/(32*BT))), z = direction
#ifdef ASR_LSTM_STAMPS
#define STAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i) do {} while (0)
#endif

// One time step of the forward recurrence for a [32*BT batch] x [32 hidden]
// tile (all four gates).  KS = H/16 k-steps (compile time).  Wave g owns gate
// g: its W_hh fragments (KS x 16 B per lane) are loaded straight to registers,
// the h_{t-1} tile — needed by all four waves — is staged once through LDS.
// BT = 2 halves the W_hh re-reads per step (the kernel is bound by operand
// fetch from L2 / Infinity Cache, not by the MFMAs).
template <int KS, int BT, int GXB>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmFwdParams p) {
#ifdef ASR_LSTM_STAMPS
    unsigned long long stamp[4];
    const unsigned long long stamp_start = __builtin_amdgcn_s_memtime();
#endif
    __shared__ __attribute__((aligned(16))) __bf16 a_lds[BT * KS * 512];
    __shared__ float g_lds[BT][4][32][33];
    const int H = p.H, B = p.B;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32 * BT, dir = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = dir == 0 ? p.step : p.T - 1 - p.step;
    const int Bp = (B + 63) & ~63;
    const __bf16 *hprev = p.hbuf + ((size_t)(p.step & 1) * 2 + dir) * Bp * H;
    __bf16 *hnext = p.hbuf + ((size_t)((p.step + 1) & 1) * 2 + dir) * Bp * H;

    // ---- W_hh fragments of this wave's gate (vmcnt retires in order: keep the
    // L2-resident operand loads ahead of the HBM loads of the pointwise phase)
    bf16x8 fb[KS];
    {
        const __bf16 *bp = p.whh + ((size_t)(dir * 4 + wave) * H * H) +
                           ((size_t)blockIdx.x * KS * 64 + lane) * 8;
#pragma unroll
        for (int k = 0; k < KS; ++k) fb[k] = *reinterpret_cast<const bf16x8 *>(bp + k * 512);
    }
    // ---- h_{t-1} tile(s): BT*KS KiB, contiguous in fragment-major order
    {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(
            hprev + (size_t)blockIdx.y * BT * KS * 512);
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(a_lds);
        constexpr int CH = BT * KS * 64;            // 16-byte chunks
        bf16x8 tmp[(CH + 255) / 256];
#pragma unroll
        for (int i = 0; i < (CH + 255) / 256; ++i) {
            const int c = i * 256 + threadIdx.x;
            if (c < CH) tmp[i] = src[c];
        }
#pragma unroll
        for (int i = 0; i < (CH + 255) / 256; ++i) {
            const int c = i * 256 + threadIdx.x;
            if (c < CH) dst[c] = tmp[i];
        }
    }
    STAMP(0);

    // ---- operands of the pointwise phase: issued now, consumed after the GEMM
    constexpr int NE = 4 * BT;
    float pgx[NE][4], pc[NE];
    bool pact[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int b = b0 + (idx >> 5), j = j0 + (idx & 31);
        const int bc = b < B ? b : B - 1;
        pact[e] = b < B && t < p.lens[bc];
        const size_t gxo = (((size_t)t * B + bc) * 2 + dir) * 4 * H + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) pgx[e][g] = load_gx<GXB>(p.gx, gxo + (size_t)g * H);
        pc[e] = p.cbuf[((size_t)dir * B + bc) * H + j];
    }
    __syncthreads();

    // ---- gate pre-activation tiles
    {
        f32x16 acc[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[bt][i] = 0.f;
        const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt)
                acc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[(bt * KS + k) * 64], fb[k],
                                                                  acc[bt], 0, 0, 0);
        STAMP(1);
        const int col = lane & 31;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                g_lds[bt][wave][row][col] = acc[bt][i];
            }
    }
    __syncthreads();
    STAMP(2);

    // ---- pointwise cell update: BT*1024 (b, j) elements over 256 threads
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int rowg = idx >> 5, col = idx & 31;
        const int bt = rowg >> 5, row = rowg & 31;
        const int b = b0 + rowg, j = j0 + col;
        if (b < B) {
            const size_t sidx = ((size_t)dir * B + b) * H + j;
            float *yo = p.y + (((size_t)t * B + b) * 2 + dir) * H + j;
            __bf16 *ybo = p.ybf + (((size_t)dir * (p.T + 2) + t + 1) * B + b) * H + j;
            const size_t csv = (((size_t)t * 2 + dir) * B + b) * H + j;
            if (pact[e]) {
                float pre[4], gt[4], c, h;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = g_lds[bt][g][row][col] + pgx[e][g];
                lstm_cell_fwd(pre, pc[e], gt, c, h);
                p.cbuf[sidx] = c;
                hnext[frag_off(b, j, KS)] = (__bf16)h;
                if (p.y) *yo = h;
                *ybo = (__bf16)h;
                p.gates[csv] = pack_gates(gt);
                p.csave[csv] = c;
            } else {
                hnext[frag_off(b, j, KS)] = hprev[frag_off(b, j, KS)];
                if (p.y) *yo = 0.f;
                *ybo = (__bf16)0.f;
                p.csave[csv] = pc[e];
            }
        }
    }
    STAMP(3);
#ifdef ASR_LSTM_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && b0 < B) {
        float *o = p.y + (((size_t)t * B + b0) * 2 + dir) * H + j0;
        o[0] = (float)(stamp[0] - stamp_start);
        for (int i = 1; i < 4; ++i) o[i] = (float)(stamp[i] - stamp[i - 1]);
    }
#endif
}

struct LstmBwdParams {
    const float *dy;        // [T,B,2,H] gradient w.r.t. the per-direction outputs, or [T,B,H]
    int dy_shared;          // 1: one gradient for both directions (the directions are summed);
                            // 2: that gradient as the sum of two planes [2][T,B,H] (the `dx`
                            //    planes of the layer above; persistent kernels only)
    const __bf16 *wihT;     // fused input gradient (lstm_bwd_dx_kernel, F == H): fragment-major pack
                            // of W_ihᵀ [2 dir][H rows (input feature)][4H cols]
    float *dx;              // ... output [2 dir][T,B,H]: dgates_dir · W_ih_dir per direction
    const __bf16 *whhT;     // fragment-major pack of W_hhᵀ: [2 dir][H rows][4H cols]
    const int32_t *lens;
    int T, B, H;
    const u32x2 *gates;     // [T,2,B,H] records of 4 bf16 (i,f,g,o)
    const float *csave;     // [T,2,B,H]
    __bf16 *dgbuf;          // [2 pingpong][2 dir] fragment-major [Bp x 4H]: dgates of the previous step
    float *dcbuf;           // [2 dir][B][H] carried dL/dc
    __bf16 *dgates;         // [T,B,2,4H] pre-activation gate gradients (output, bf16: GEMM operand)
    int step;
};

// One time step of the backward recurrence for a [32*BT batch] x [32 hidden]
// tile: dh_rec[b][j] = sum_k dgates_prev[b][k] * W_hh[k][j] with K = 4H split
// over the four waves (wave w takes the columns of gate w), then the LSTM cell
// backward in fp32.
template <int KS, int BT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmBwdParams p) {
    __shared__ float part[BT][4][32][33];
    const int H = p.H, B = p.B, H4 = 4 * p.H;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32 * BT, dir = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the backward scan visits frames in the opposite order of the forward one
    const int t = dir == 0 ? p.T - 1 - p.step : p.step;
    const int Bp = (B + 63) & ~63;
    constexpr int KS4 = 4 * KS;
    const __bf16 *dgprev = p.dgbuf + ((size_t)(p.step & 1) * 2 + dir) * Bp * H4;
    __bf16 *dgnext = p.dgbuf + ((size_t)((p.step + 1) & 1) * 2 + dir) * Bp * H4;

    // ---- operands: W_hhᵀ slice of this wave (registers) and the dgates tiles
    bf16x8 fb[KS], fa[BT][KS];
    {
        const __bf16 *bp = p.whhT + (size_t)dir * H * H4 +
                           (((size_t)blockIdx.x * KS4 + (size_t)wave * KS) * 64 + lane) * 8;
#pragma unroll
        for (int k = 0; k < KS; ++k) fb[k] = *reinterpret_cast<const bf16x8 *>(bp + k * 512);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const __bf16 *ap = dgprev + ((((size_t)blockIdx.y * BT + bt) * KS4 +
                                          (size_t)wave * KS) * 64 + lane) * 8;
#pragma unroll
            for (int k = 0; k < KS; ++k)
                fa[bt][k] = *reinterpret_cast<const bf16x8 *>(ap + k * 512);
        }
    }

    // ---- operands of the pointwise phase: issued now, consumed after the GEMM
    constexpr int NE = 4 * BT;
    float pg[NE][4], pcs[NE], pcp[NE], pdy[NE], pdc[NE];
    bool pact[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int b = b0 + (idx >> 5), j = j0 + (idx & 31);
        const int bc = b < B ? b : B - 1;
        const int len = p.lens[bc];
        pact[e] = b < B && t < len;
        unpack_gates(p.gates[(((size_t)t * 2 + dir) * B + bc) * H + j], pg[e]);
        pcs[e] = p.csave[(((size_t)t * 2 + dir) * B + bc) * H + j];
        const int tp = dir == 0 ? t - 1 : t + 1;
        const int tpc = tp < 0 ? 0 : (tp >= p.T ? p.T - 1 : tp);
        const float cpv = p.csave[(((size_t)tpc * 2 + dir) * B + bc) * H + j];
        pcp[e] = (tp >= 0 && tp < len) ? cpv : 0.f;
        pdy[e] = p.dy[p.dy_shared ? ((size_t)t * B + bc) * H + j : (((size_t)t * B + bc) * 2 + dir) * H + j];
        pdc[e] = p.dcbuf[((size_t)dir * B + bc) * H + j];
    }

    {
        const int col = lane & 31;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bt][k], fb[k], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                part[bt][wave][row][col] = acc[i];
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = e * 256 + threadIdx.x;
        const int rowg = idx >> 5, col = idx & 31;
        const int bt = rowg >> 5, row = rowg & 31;
        const int b = b0 + rowg, j = j0 + col;
        if (b < B) {
            const size_t sidx = ((size_t)dir * B + b) * H + j;
            __bf16 *dgo = p.dgates + (((size_t)t * B + b) * 2 + dir) * H4 + j;
            if (pact[e]) {
                const float dh = pdy[e] + (part[bt][0][row][col] + part[bt][1][row][col]) +
                                 (part[bt][2][row][col] + part[bt][3][row][col]);
                float d[4], dcout;
                lstm_cell_bwd(pg[e], pcs[e], pcp[e], dh, pdc[e], d, dcout);   // pcp: cell state the step started from
                const float d_i = d[0], d_f = d[1], d_g = d[2], d_o = d[3];
                p.dcbuf[sidx] = dcout;
                dgo[0] = (__bf16)d_i; dgo[H] = (__bf16)d_f;
                dgo[2 * H] = (__bf16)d_g; dgo[3 * H] = (__bf16)d_o;
                dgnext[frag_off(b, j, KS4)] = (__bf16)d_i;
                dgnext[frag_off(b, H + j, KS4)] = (__bf16)d_f;
                dgnext[frag_off(b, 2 * H + j, KS4)] = (__bf16)d_g;
                dgnext[frag_off(b, 3 * H + j, KS4)] = (__bf16)d_o;
            } else {
                // no gradient reaches a padding frame; the carried state gradient
                // restarts from zero (forward: beyond the end; reverse: before the start)
                p.dcbuf[sidx] = 0.f;
                dgo[0] = (__bf16)0.f; dgo[H] = (__bf16)0.f;
                dgo[2 * H] = (__bf16)0.f; dgo[3 * H] = (__bf16)0.f;
                dgnext[frag_off(b, j, KS4)] = (__bf16)0.f;
                dgnext[frag_off(b, H + j, KS4)] = (__bf16)0.f;
                dgnext[frag_off(b, 2 * H + j, KS4)] = (__bf16)0.f;
                dgnext[frag_off(b, 3 * H + j, KS4)] = (__bf16)0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent recurrence: ONE launch walks all T steps.  A workgroup (512
// threads) owns a [32 batch x 64 hidden] tile for the whole sequence: its W_hh
// fragments (KS x 16 B per lane per wave) and the cell state never leave
// registers; only h_t (forward) / dgates_t (backward) crosses workgroups, once
// per step, inside a TEAM = the H/64 workgroups of one (direction, batch tile).
// Hand-off (MI355X_MICROARCH.md § visibility, sc1 table rows 1/3): the tile is
// staged in LDS in fragment order and stored as whole 1 KiB blocks, one
// `buffer_store_dwordx4 sc1` (write-through) per wave; every wave drains
// vmcnt(0), the workgroup meets at a barrier, ONE lane adds 1 to the team
// counter (agent scope); consumers poll that counter with an sc1 load from one
// lane, pass a workgroup barrier, then read the tile with `buffer_load_dwordx4
// sc1` only.  The grid never exceeds one workgroup per CU (LDS request > 80 KiB
// and grid <= #CUs, the batch is cut into several launches if needed), so every
// team mate is resident; every spin is bounded and a timeout poisons the
// outputs with NaN instead of hanging.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define ASR_SC1 16

__device__ __forceinline__ bool team_wait(unsigned *ctr, unsigned target, unsigned limit,
                                          unsigned *err) {
    for (unsigned it = 0; it < limit; ++it) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target)
            return true;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

#ifdef ASR_LSTM_STAMPS
#define PSTAMP_DECL unsigned long long pst[6] = {0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime()
#define PSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                       pst[i] += now_ - plast; plast = now_; } while (0)
#else
#define PSTAMP_DECL do {} while (0)
#define PSTAMP(i) do {} while (0)
#endif

// LDS-DMA: 64 lanes x 16 bytes from a bounds-checked raw buffer straight to LDS at
// `lds_byte` + lane * 16 (out-of-range lanes write zeros).  Inline asm: hipcc would order
// every later LDS read behind it with s_waitcnt vmcnt(0) (cdna_hip_programming.md §5.7);
// the kernel waits where it needs the data.
typedef __attribute__((ext_vector_type(4))) int rsrc_words;
__device__ __forceinline__ rsrc_words raw_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    rsrc_words r;            // (readfirstlane: the descriptor must sit in SGPRs for the asm)
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16_s(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void dma16_sc1(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// four of them, 1 KiB apart on BOTH sides (the instruction offset moves the LDS and the
// global address alike): one M0 set-up for four k-steps of a fragment-major tile
__device__ __forceinline__ void dma16x4_sc1(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 sc1 lds\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen offset:2048 sc1 lds\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen offset:3072 sc1 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// 64 lanes x 4 bytes
__device__ __forceinline__ void dma4_s(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *ptr) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const void *)ptr);
}

// acc += A[32 x 16*(K1-K0)] · B over the k-steps K0..K1-1: the A fragments (1 KiB apart in
// LDS, fragment-major) are read D k-steps ahead into a rotating register set.  Written as
// one dependent ds_read -> MFMA pair per k-step the chain pays the LDS latency (~100
// cycles) 20 times per phase: measured 1998 cycles for 20 MFMAs that need 640 in the pipe.
template <int K0, int K1, int D, int KS>
__device__ __forceinline__ f32x16 mfma_chain(const bf16x8 *al, const bf16x8 (&fb)[KS], f32x16 acc) {
    bf16x8 a[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (K0 + d < K1) a[d] = al[(K0 + d) * 64];
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k - K0) % D], fb[k], acc, 0, 0, 0);
        if (k + D < K1) a[(k - K0) % D] = al[(k + D) * 64];
    }
    // keep that order: the scheduler otherwise sinks every read next to its MFMA again
#pragma unroll
    for (int d = 0; d < D; ++d) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k + D < K1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    return acc;
}

struct LstmTeamCtl {
    unsigned *ctr;          // [2 dir][nbt] counters, 32 words (128 B) apart, zeroed per call
    unsigned *err;          // timeout word
    unsigned spin_limit;
    int bt0, nbt;           // first batch tile of this launch, batch tiles in total
    int xcd_teams;          // 0: grid (jt, batch tile, dir); else teams of the XCD-affine 1-D grid
    size_t rows;            // rows of one hbuf / dgbuf plane (>= 32 * nbt)
};

#define ASR_GLDS_BYTES (4 * 32 * 65 * 4)

// NE = batch rows of the tile / 8 (tiles of 16, 24 or 32 rows: fewer rows per workgroup
// = more workgroups; the host picks the smallest tile whose grid still fits one
// workgroup per CU).  Rows >= 8*NE of the 32-row MFMA tile are padding.
//
// XF > 0 (= the input size F / 16: the projection's k-steps) fuses the input projection: instead of reading x·W_ihᵀ (`gx`, the output of a
// [T·B, F] x [F, 8H] library GEMM: 0.37 ms and 2 GB of HBM traffic per layer at B=576) the
// workgroup keeps its W_ih slice in registers next to W_hh (F == H, or the first layer's
// F = 352 behind the conv front-end: 22 k-steps) and multiplies the x_t
// tile itself — x_t does not depend on the recurrence, so waves 4-7 do it while the
// workgroup waits for the team's counter and waves 0-3 while the hand-off tile is in
// flight: both windows (911 / 1036 cycles at B=512) were idle.  The sum x_t·W_ih + h·W_hh
// is accumulated in fp32 in one MFMA chain (the bf16 rounding of `gx` is gone).
// SUM: the launch also writes p.xsum (see LstmFwdParams) — its own instantiation, so that the
// other launches keep their register allocation (the XF kernels sit at 247-256 VGPRs)
template <int KS, int NE, int GXB, int XF, int SUM = 0>
__global__ __launch_bounds__(512) void lstm_fwd_persist_kernel(LstmFwdParams p, LstmTeamCtl ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *a_lds = reinterpret_cast<__bf16 *>(smem);                       // KS KiB
    float (*g_lds)[32][65] = reinterpret_cast<float (*)[32][65]>(smem + KS * 1024);
    __bf16 *h_lds = reinterpret_cast<__bf16 *>(smem + KS * 1024 + ASR_GLDS_BYTES);   // 4 KiB
    __bf16 *x_lds = reinterpret_cast<__bf16 *>(smem + KS * 1024 + ASR_GLDS_BYTES + 4096);   // XF KiB
    __bf16 *o_lds = reinterpret_cast<__bf16 *>(smem + KS * 1024 + ASR_GLDS_BYTES + 4096 + XF * 1024);   // 2 x 4 KiB (SUM)
    constexpr int KX = XF > 0 ? XF : 1;
    __shared__ int dead_s;
    const int H = p.H, B = p.B, T = p.T;
    int jt = blockIdx.x, btile = blockIdx.y + ctl.bt0, dir = blockIdx.z;
    if (ctl.xcd_teams) {
        // 1-D grid, team = blockIdx.x % 8 + 8 * (slot / njt): the H/64 workgroups that
        // exchange h_t / dgates_t every step have equal blockIdx % 8, i.e. sit on one XCD
        // under the observed round-robin placement (speed only — the hand-off protocol
        // does not depend on it); surplus workgroups of the padded grid leave at once
        const int slot = blockIdx.x >> 3, team = (slot / (p.H / 64)) * 8 + (blockIdx.x & 7);
        if (team >= ctl.xcd_teams) return;
        jt = slot % (p.H / 64);
        btile = (team >> 1) + ctl.bt0;
        dir = team & 1;
    }
    const int j0 = jt * 64, b0 = btile * (8 * NE), njt = H / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int gate = wave >> 1, js = wave & 1;
    const size_t Bp = ctl.rows;              // rows of an hbuf / dgbuf plane (32 per batch tile)
    unsigned *myctr = ctl.ctr + ((size_t)dir * ctl.nbt + btile) * 32;
    if (tid == 0) dead_s = 0;
    for (int i = tid; i < 2048; i += 512) h_lds[i] = (__bf16)0.f;       // padding rows stay 0
    for (int i = tid; i < KS * 64; i += 512)                            // (the tile's too: never loaded)
        reinterpret_cast<u32x4 *>(a_lds)[i] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (XF)
        for (int i = tid; i < KX * 64; i += 512)
            reinterpret_cast<u32x4 *>(x_lds)[i] = u32x4{0u, 0u, 0u, 0u};

    bf16x8 fb[KS], fx[KX];
    {
        const size_t wo = ((size_t)(dir * 4 + gate) * H * H) + ((size_t)(2 * jt + js) * KS * 64 + lane) * 8;
#pragma unroll
        for (int k = 0; k < KS; ++k) fb[k] = *reinterpret_cast<const bf16x8 *>(p.whh + wo + k * 512);
        if constexpr (XF) {       // W_ih pack: [2*4][H rows][16*KX cols], fragment-major
            const size_t wox = ((size_t)(dir * 4 + gate) * H * (16 * KX)) + ((size_t)(2 * jt + js) * KX * 64 + lane) * 8;
#pragma unroll
            for (int k = 0; k < KX; ++k) fx[k] = *reinterpret_cast<const bf16x8 *>(p.wih + wox + k * 512);
        }
    }
    const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc(
        p.hbuf, 0, (int)(2 * 2 * Bp * H * 2), 0x00020000);
    const rsrc_words hresD = raw_rsrc(p.hbuf, (unsigned)(2 * 2 * Bp * H * 2));

    // the thread's four (row, col) elements: row = e*8 + wave, col = lane
    const int col = lane;
    float c[NE];
    __bf16 hq[NE];
    int len[NE];
    // Every per-step global access of the pointwise part goes through a raw buffer: the
    // lane's column as the only per-lane offset (4, 2 or 8 bytes per column), the row and
    // frame parts — wave-uniform: a wave's rows are b0 + 8 e + wave — in SGPRs; no 64-bit
    // address arithmetic in the step loop, and masked rows simply carry an out-of-range lane
    // offset (stores dropped).  The host checks every tensor is < 4 GiB.
    typedef unsigned int u32;
    constexpr u32 OOBV = 0xFFFFFFFFu;
    constexpr u32 GXE = GXB ? 2u : 4u;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const u32 l4 = (u32)col * 4u, l2 = (u32)col * 2u;
    u32 sgx[NE], sy[NE], syb[NE], scs[NE];     // row parts (bytes); gate records: 2 * scs
    bool inb[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int b = b0 + e * 8 + wv;
        const int bc = b < B ? b : B - 1;
        inb[e] = b < B;
        c[e] = 0.f;
        hq[e] = (__bf16)0.f;
        len[e] = __builtin_amdgcn_readfirstlane(b < B ? p.lens[bc] : 0);
        sgx[e] = (u32)((bc * 2 + dir) * 4 * H + j0) * GXE;
        sy[e] = (u32)((b * 2 + dir) * H + j0) * 4u;
        syb[e] = (u32)(((size_t)(dir * (T + 2) + 1) * B + b) * H + j0) * 2u;
        scs[e] = (u32)((dir * B + b) * H + j0) * 4u;
    }
    const u32 fgx = (u32)B * 8u * H * GXE, fy = (u32)B * 2u * H * 4u, fyb = (u32)B * H * 2u;
    const u32 fcs = (u32)B * 2u * H * 4u, fg = (u32)B * 2u * H * 8u;
    const __amdgpu_buffer_rsrc_t gxR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.gx), 0, XF ? 0 : (int)((u32)T * fgx), 0x00020000);
    // XF: the x_t tile goes global -> LDS by LDS-DMA, no registers: the 1 KiB A fragment of
    // k-step k is lane-linear in LDS (lane = row + 32 * k-half, 16 bytes each), which is
    // exactly the destination pattern of one `buffer_load_dwordx4 ... lds`; lanes of the
    // tile's padding rows carry an out-of-range offset and write zeros.  Wave w moves
    // k-steps w, w + 8, w + 16.
    const rsrc_words xD = raw_rsrc(p.x, XF ? (unsigned)((u32)T * B * (16 * KX) * 2u) : 0u);
    const u32 fx_frame = (u32)B * (16 * KX) * 2u;
    u32 vx = 0x80000000u;
    if constexpr (XF) {
        const int xr = lane & 31, b = b0 + xr < B ? b0 + xr : B - 1;
        if (xr < 8 * NE) vx = (u32)(b * (16 * KX) + 16 * wv + 8 * (lane >> 5)) * 2u;
    }
    const int S0 = p.s_begin, S1 = p.s_end;
    constexpr bool summ = XF && SUM;
    const rsrc_words ybD = raw_rsrc(p.ybf, summ ? (unsigned)((u32)(2 * (T + 2)) * fyb) : 0u);
    const __amdgpu_buffer_rsrc_t xsR = __builtin_amdgcn_make_buffer_rsrc(
        p.xsum, 0, summ ? (int)((u32)T * fyb) : 0, 0x00020000);
    auto x_dma = [&](int tq) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (wave + 8 * i < KX)
                dma16_s(xD, (unsigned)__builtin_amdgcn_readfirstlane(
                                (int)(lds_addr(x_lds) + (unsigned)(wave + 8 * i) * 1024u)),
                        vx + (u32)i * 256u, (u32)__builtin_amdgcn_readfirstlane((int)((u32)tq * fx_frame)));
    };
    // SUM: the other direction's bf16 outputs of frame tq (written by an earlier launch), rows
    // 8 wave .. + 7 of the tile x this workgroup's 64 columns: 1 KiB per wave 0..3, lane = (row,
    // 16-byte chunk), into buffer `par` of o_lds.  Fetched one step ahead like x: these bytes come
    // from HBM, and a DMA issued with the hand-off tile put their 2 us on every step's
    // critical path (measured: 4.65 instead of 3.9 us per step).
    auto o_dma = [&](int tq, int par) {
        if (wave < 4) {
            int ln = lane;      // (an opaque copy: hoisted out of the loop these offsets cost registers)
            asm volatile("" : "+v"(ln));
            const int orow = 8 * wv + (ln >> 3);
            const u32 vo = orow < 8 * NE && b0 + orow < B
                               ? (u32)(((b0 + orow) * H + j0) * 2 + (ln & 7) * 16) : 0x80000000u;
            dma16_s(ybD, (unsigned)__builtin_amdgcn_readfirstlane(
                             (int)(lds_addr(o_lds) + (unsigned)par * 4096u + (unsigned)wv * 1024u)), vo,
                    (u32)__builtin_amdgcn_readfirstlane((int)((u32)((1 - dir) * (T + 2) + tq + 1) * fyb)));
        }
    };
    // y == null: zero records, every fp32 output store is dropped by the bounds check
    const __amdgpu_buffer_rsrc_t yR = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y ? (int)((u32)T * fy) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ybR = __builtin_amdgcn_make_buffer_rsrc(
        p.ybf, 0, (int)((u32)(2 * (T + 2)) * fyb), 0x00020000);
    const __amdgpu_buffer_rsrc_t csR = __builtin_amdgcn_make_buffer_rsrc(p.csave, 0, (int)((u32)T * fcs), 0x00020000);
    const __amdgpu_buffer_rsrc_t gR = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)((u32)T * fg), 0x00020000);
    auto ld_gx = [&](u32 voff, u32 soff) -> float {
        if constexpr (GXB)
            return (float)__builtin_bit_cast(__bf16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(gxR, voff, soff, 0));
        else
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gxR, voff, soff, 0));
    };

    // Software pipeline: the x·W_ih terms of step s+1 are fetched during step s
    // (behind the hand-off tile in the wave's in-order vmcnt queue, never in
    // front of the team poll); the outputs nobody inside the launch reads are
    // stored after the team signal, under the hand-off latency.
    if (S0 > 0) {
        // pick the state up where the previous launch left it: c_{t-1} from csave, this
        // workgroup's own part of h_{t-1} from the hand-off buffer (the same bytes h_lds held)
        const int tp = dir == 0 ? S0 - 1 : T - S0;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            c[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                csR, inb[e] ? l4 : OOBV, (u32)tp * fcs + scs[e], 0));
        __syncthreads();                // the zero fill of h_lds above
        if (tid < 256) {
            const unsigned off = (unsigned)(((((size_t)(S0 & 1) * 2 + dir) * Bp * H) +
                                             ((size_t)btile * KS + 4 * jt + (tid >> 6)) * 512 + (tid & 63) * 8) * 2);
            reinterpret_cast<u32x4 *>(h_lds)[tid] = __builtin_amdgcn_raw_buffer_load_b128(
                hres, (tid & 31) < 8 * NE ? off : 0xFFFFFFFFu, 0, ASR_SC1);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = e * 8 + wave;
            hq[e] = h_lds[(((col >> 4) * 64) + row + 32 * ((col >> 3) & 1)) * 8 + (col & 7)];
        }
    }
    float pgx[XF ? 1 : NE][4];
    {
        const int t0 = dir == 0 ? S0 : T - 1 - S0;
        if constexpr (XF) {
            __syncthreads();            // the zero fill above
            x_dma(t0);
            if constexpr (summ) o_dma(t0, S0 & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) pgx[e][g] = ld_gx((u32)col * GXE, sgx[e] + (u32)t0 * fgx + (u32)g * H * GXE);
        }
    }
    // A fragments in flight (registers are short with two weight slices: read one k-step ahead
    // there — one by one the chain pays the LDS latency per MFMA, 3x the window it has to fit).
    // The first layer's variant (22 input k-steps) writes no fp32 y: it is never the last layer
    // of a stack.
    constexpr int AD = XF ? 2 : 4;
    constexpr bool NOY = XF > KS;
    f32x16 accx;                 // XF: x_t · W_ih of this wave's (gate, column half)
    auto x_mfma = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) accx[i] = 0.f;
        const bf16x8 *xl = reinterpret_cast<const bf16x8 *>(x_lds) + lane;
        if constexpr (XF) accx = mfma_chain<0, KX, AD>(xl, fx, accx);
    };
    u32x2 sog[NE] = {};
    float soh[NE] = {}, sc[NE] = {};
    __bf16 shq[NE] = {};                 // NOY: the bf16 output of the step (0 on padding frames)
    bool sact[NE] = {};
    int st = 0, st_par = 0;
    // live == false (the first step has nothing to store yet): every offset out of range,
    // so the call has the same VMEM count on every step (counted vmcnt waits around it)
    auto bulk_store = [&](bool live) {
        const u32 ust = (u32)st;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool on = live && inb[e];
            if constexpr (!NOY)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, soh[e]), yR,
                                                      on ? l4 : OOBV, ust * fy + sy[e], 0);
            __builtin_amdgcn_raw_buffer_store_b16(
                (short)__builtin_bit_cast(unsigned short, NOY ? shq[e] : (__bf16)soh[e]), ybR,
                on ? l2 : OOBV, ust * fyb + syb[e], 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sc[e]), csR,
                                                  on ? l4 : OOBV, ust * fcs + scs[e], 0);
            __builtin_amdgcn_raw_buffer_store_b64(sog[e], gR,
                                                  on && sact[e] ? l4 * 2u : OOBV, ust * fg + scs[e] * 2u, 0);
        }
    };
    // SUM: own output + the other direction's (buffer st_par of o_lds), issued BEHIND the team
    // signal: its LDS reads would otherwise sit between the hand-off store and the signal
    auto sum_store = [&]() {
        const u32 ust = (u32)st;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool on = inb[e];
            int cl = col;
            asm volatile("" : "+v"(cl));
            const __bf16 xs = (__bf16)((float)shq[e] + (float)o_lds[st_par * 2048 + (e * 8 + wave) * 64 + cl]);
            __builtin_amdgcn_raw_buffer_store_b16(
                (short)__builtin_bit_cast(unsigned short, xs), xsR, on ? l2 : OOBV,
                ust * fyb + (u32)__builtin_amdgcn_readfirstlane(((b0 + e * 8 + wv) * H + j0) * 2), 0);
        }
    };

    if constexpr (XF) {
        // Everything the prologue loaded must have landed BEFORE the loop: the compiler does
        // not see the LDS-DMA inside the asm, so a wait it places inside the loop for one of
        // these (first use of a weight fragment / a length) is counted without them and
        // degenerates to vmcnt(0) right behind the x_{t+1} DMA — the MFMA phase then waits
        // for the DMA (measured: 1992 -> 4091 cycles).  Naming the registers as asm operands
        // makes it wait here.
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            asm volatile("" : "+v"(fb[k]));
        }
#pragma unroll
        for (int k = 0; k < KX; ++k) asm volatile("" : "+v"(fx[k]));
#pragma unroll
        for (int e = 0; e < NE; ++e) asm volatile("" : "+v"(len[e]));
    }
    PSTAMP_DECL;
    for (int step = S0; step < S1; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        if constexpr (XF) if (wave >= 4) x_mfma();          // under the team wait
        if (step > 0 && tid == 0 && !dead_s) {
            if (!team_wait(myctr, (unsigned)(njt * step), ctl.spin_limit, ctl.err)) dead_s = 1;
        }
        PSTAMP(0);
        __syncthreads();
        // ---- h_{t-1} tile of the team: the KS - 4 KiB of the OTHER workgroups come over
        // sc1 loads; this workgroup's own 4 KiB are still in h_lds (staged there for the
        // store of the previous step: the same bytes) and are copied LDS -> LDS
        // (only the 2 * 8*NE chunks of a k-step that carry real batch rows move; the
        // padding rows of the 32-row MFMA tile stay zero in LDS)
        constexpr int RL = 8 * NE, CPK = 2 * RL;
        constexpr int CHO = (KS - 4) * CPK, NI = XF ? 0 : (CHO + 511) / 512;
        u32x4 tmp[NI > 0 ? NI : 1];
        if constexpr (XF) {
            // with two weight slices in registers there is none left to stage the tile in: it
            // comes global -> LDS by LDS-DMA (the hand-off buffer is fragment-major, 1 KiB per
            // k-step, lane-linear on both sides; padding rows read out of range = zeros)
            const unsigned base = (unsigned)((((size_t)(step & 1) * 2 + dir) * Bp * H +
                                              (size_t)btile * KS * 512) * 2);
            const u32 voffT = (lane & 31) < RL ? (u32)lane * 16u : 0x80000000u;
            for (int ko = wv; ko < KS - 4; ko += 8) {
                const int kk = ko < 4 * jt ? ko : ko + 4;
                dma16_sc1(hresD, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr(a_lds) + (unsigned)kk * 1024u)),
                          voffT, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + (unsigned)kk * 1024u)));
            }
        }
        if constexpr (NI > 0) {
            const unsigned base = (unsigned)((((size_t)(step & 1) * 2 + dir) * Bp * H +
                                              (size_t)btile * KS * 512) * 2);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int co = i * 512 + tid;       // lanes past the tile re-read its last chunk
                const int cc = co < CHO ? co : CHO - 1;
                const int ko = cc / CPK, q = cc - ko * CPK, k = ko < 4 * jt ? ko : ko + 4;
                tmp[i] = __builtin_amdgcn_raw_buffer_load_b128(
                    hres, base + (k * 64 + (q < RL ? q : q + (32 - RL))) * 16, 0, ASR_SC1);
            }
        }
        if (tid < 256)
            reinterpret_cast<u32x4 *>(a_lds)[(4 * jt + (tid >> 6)) * 64 + (tid & 63)] =
                reinterpret_cast<const u32x4 *>(h_lds)[tid];
        if constexpr (XF) {
            if (wave < 4) x_mfma();                         // under the hand-off tile's flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tile
        }
        if constexpr (NI > 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int co = i * 512 + tid;
                const int ko = co / CPK, q = co - ko * CPK, k = ko < 4 * jt ? ko : ko + 4;
                if (co < CHO)
                    reinterpret_cast<u32x4 *>(a_lds)[k * 64 + (q < RL ? q : q + (32 - RL))] = tmp[i];
            }
        }
        __syncthreads();
        PSTAMP(1);
        float ngx[XF ? 1 : NE][4];
        {
            const int sn = step + 1 < T ? step + 1 : step;
            const int tn = dir == 0 ? sn : T - 1 - sn;
            if constexpr (XF) {
                // x_{t+1}: every read of x_lds for this step lies before the tile barrier above;
                // landed at the s_waitcnt vmcnt(0) in front of the signal barrier below
                // (32-row tiles keep two weight fragments in scratch; their reloads inside the
                // MFMA loop would wait for this DMA: issued behind the loop there)
                if constexpr (NE < 4) x_dma(tn);
                if constexpr (summ) o_dma(tn, (step + 1) & 1);      // (read in the cell phase of step + 1)
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        ngx[e][g] = ld_gx((u32)col * GXE, sgx[e] + (u32)tn * fgx + (u32)g * H * GXE);
            }
        }
        {
            f32x16 acc;
            if constexpr (XF) acc = accx;
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + lane;
            acc = mfma_chain<0, KS, AD>(al, fb, acc);
            const int c32 = lane & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                g_lds[gate][row][js * 32 + c32] = acc[i];
            }
        }
        if constexpr (XF && NE == 4) {
            const int sn = step + 1 < T ? step + 1 : step;
            x_dma(dir == 0 ? sn : T - 1 - sn);
        }
        __syncthreads();
        PSTAMP(2);
        const bool dead = dead_s != 0;
        st = t;
        st_par = step & 1;
        auto cell_one = [&](int e) {
            const int row = e * 8 + wave;
            if (sact[e]) {
                float pre[4], cn, og[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = XF ? g_lds[g][row][col] : g_lds[g][row][col] + pgx[XF ? 0 : e][g];
                lstm_cell_fwd(pre, c[e], og, cn, soh[e]);
                sog[e] = pack_gates(og);
                c[e] = cn;
                if (dead) soh[e] = __builtin_nanf("");
                hq[e] = (__bf16)soh[e];
                shq[e] = hq[e];
            } else {
                soh[e] = 0.f;
                shq[e] = (__bf16)0.f;
            }
        };
#pragma unroll
        for (int e = 0; e < NE; ++e) sact[e] = t < len[e];
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            if (e + 1 < NE && sact[e] && sact[e + 1]) {       // (lengths are wave-uniform)
                const int row = e * 8 + wave;
                f32x2 pre[4], cn, hn, og[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pre[g] = f32x2{g_lds[g][row][col], g_lds[g][row + 8][col]};
                    if constexpr (!XF) pre[g] = pre[g] + f32x2{pgx[XF ? 0 : e][g], pgx[XF ? 0 : e + 1][g]};
                }
                lstm_cell_fwd2(pre, f32x2{c[e], c[e + 1]}, og, cn, hn);
                const float og0[4] = {og[0].x, og[1].x, og[2].x, og[3].x}, og1[4] = {og[0].y, og[1].y, og[2].y, og[3].y};
                sog[e] = pack_gates(og0);
                sog[e + 1] = pack_gates(og1);
                c[e] = cn.x; c[e + 1] = cn.y;
                soh[e] = dead ? __builtin_nanf("") : hn.x;
                soh[e + 1] = dead ? __builtin_nanf("") : hn.y;
                hq[e] = (__bf16)soh[e]; hq[e + 1] = (__bf16)soh[e + 1];
                shq[e] = hq[e]; shq[e + 1] = hq[e + 1];
            } else {
                cell_one(e);
                if (e + 1 < NE) cell_one(e + 1);
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = e * 8 + wave;
            sc[e] = c[e];
            h_lds[(((col >> 4) * 64) + row + 32 * ((col >> 3) & 1)) * 8 + (col & 7)] = hq[e];
        }
        __syncthreads();
        if (wave < 4) {
            const u32x4 v = reinterpret_cast<const u32x4 *>(h_lds)[wave * 64 + lane];
            const unsigned off = (unsigned)(((((size_t)((step + 1) & 1) * 2 + dir) * Bp * H) +
                                             ((size_t)btile * KS + 4 * jt + wave) * 512 + lane * 8) * 2);
            // padding rows are not handed over (nobody loads them)
            __builtin_amdgcn_raw_buffer_store_b128(v, hres, (lane & 31) < 8 * NE ? off : 0xFFFFFFFFu,
                                                   0, ASR_SC1);
        }
        PSTAMP(3);
        // outputs nobody inside this launch reads: issued BEHIND the hand-off store and not
        // waited for — the counted wait retires everything up to the hand-off store (VMEM
        // retires in issue order), so its write-through latency (613 cycles of pure waiting)
        // runs under the issue of these stores.  (Issuing them behind the NEXT step's tile
        // loads instead, as the backward kernel does, was measured 1 % slower here.)
        bulk_store(true);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NOY ? 3 : 4) * NE) : "memory");
        __syncthreads();
        PSTAMP(4);
        if (tid == 0) __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (summ) sum_store();
        if constexpr (!XF) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) pgx[e][g] = ngx[e][g];
        }
        PSTAMP(5);
    }
#ifdef ASR_LSTM_STAMPS
    __syncthreads();
    if (tid == 0 && b0 < B)
        for (int i = 0; i < 6; ++i) p.y[(((size_t)0 * B + b0) * 2 + dir) * H + j0 + i] = (float)pst[i] / T;
#endif
}

template <int KS, int NE>
__global__ __launch_bounds__(512) void lstm_bwd_persist_kernel(LstmBwdParams p, LstmTeamCtl ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS4 = 4 * KS;
    __bf16 *a_lds = reinterpret_cast<__bf16 *>(smem);                       // 4*KS KiB
    float (*part)[32][65] = reinterpret_cast<float (*)[32][65]>(smem + KS4 * 1024);
    __bf16 *dg_lds = reinterpret_cast<__bf16 *>(smem + KS4 * 1024 + ASR_GLDS_BYTES);   // 16 KiB
    __shared__ int dead_s;
    const int H = p.H, B = p.B, T = p.T, H4 = 4 * p.H;
    int jt = blockIdx.x, btile = blockIdx.y + ctl.bt0, dir = blockIdx.z;
    if (ctl.xcd_teams) {
        // 1-D grid, team = blockIdx.x % 8 + 8 * (slot / njt): the H/64 workgroups that
        // exchange h_t / dgates_t every step have equal blockIdx % 8, i.e. sit on one XCD
        // under the observed round-robin placement (speed only — the hand-off protocol
        // does not depend on it); surplus workgroups of the padded grid leave at once
        const int slot = blockIdx.x >> 3, team = (slot / (p.H / 64)) * 8 + (blockIdx.x & 7);
        if (team >= ctl.xcd_teams) return;
        jt = slot % (p.H / 64);
        btile = (team >> 1) + ctl.bt0;
        dir = team & 1;
    }
    const int j0 = jt * 64, b0 = btile * (8 * NE), njt = H / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = wave >> 1, js = wave & 1;
    const size_t Bp = ctl.rows;              // rows of an hbuf / dgbuf plane (32 per batch tile)
    unsigned *myctr = ctl.ctr + ((size_t)dir * ctl.nbt + btile) * 32;
    if (tid == 0) dead_s = 0;
    for (int i = tid; i < 8192; i += 512) dg_lds[i] = (__bf16)0.f;      // padding rows stay 0
    for (int i = tid; i < KS4 * 64; i += 512)                           // (the tile's too: never loaded)
        reinterpret_cast<u32x4 *>(a_lds)[i] = u32x4{0u, 0u, 0u, 0u};

    bf16x8 fb[KS];
    {
        const __bf16 *bp = p.whhT + (size_t)dir * H * H4 +
                           (((size_t)(2 * jt + js) * KS4 + (size_t)kq * KS) * 64 + lane) * 8;
#pragma unroll
        for (int k = 0; k < KS; ++k) fb[k] = *reinterpret_cast<const bf16x8 *>(bp + k * 512);
    }
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(
        p.dgbuf, 0, (int)(2 * 2 * Bp * H4 * 2), 0x00020000);

    const int col = lane, j = j0 + col;
    float dcarry[NE];
    int len[NE], bcl[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int b = b0 + e * 8 + wave;
        dcarry[e] = 0.f;
        len[e] = b < B ? p.lens[b] : 0;
        bcl[e] = b < B ? b : B - 1;
    }

    // operands of the cell backward at frame t: saved gates, dy, and the cell
    // state the step started from (c of the neighbouring frame, raw); c_t itself
    // is the neighbour value fetched one step earlier
    // raw buffers, per-element 32-bit byte offsets computed once, frame offsets in SGPRs
    // (as in the forward kernel; the host checks every tensor is < 4 GiB)
    typedef unsigned int u32;
    constexpr u32 OOBV = 0xFFFFFFFFu;
    u32 vcs[NE], vdy[NE], vdg[NE];        // gate records: 2 * vcs
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int b = b0 + e * 8 + wave;
        vcs[e] = (u32)((dir * B + bcl[e]) * H + j) * 4u;
        vdy[e] = (u32)(p.dy_shared ? bcl[e] * H + j : (bcl[e] * 2 + dir) * H + j) * 4u;
        vdg[e] = b < B ? (u32)((b * 2 + dir) * H4 + j) * 2u : OOBV;
    }
    const u32 fg = (u32)B * 2u * H * 8u, fcs = (u32)B * 2u * H * 4u;
    const u32 fdy = (u32)B * H * 4u * (p.dy_shared ? 1u : 2u), fdg = (u32)B * 2u * H4 * 2u;
    // dy_shared == 2: the second plane lies T frames further on; otherwise that load is out
    // of range and returns 0
    const u32 dy2 = p.dy_shared == 2 ? (u32)T * fdy : 0x80000000u;
    const __amdgpu_buffer_rsrc_t gR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x2 *>(p.gates), 0, (int)((u32)T * fg), 0x00020000);
    const __amdgpu_buffer_rsrc_t csR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.csave), 0, (int)((u32)T * fcs), 0x00020000);
    const __amdgpu_buffer_rsrc_t dyR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.dy), 0,
        __builtin_amdgcn_readfirstlane((int)((u32)T * fdy * (p.dy_shared == 2 ? 2u : 1u))), 0x00020000);
    const __amdgpu_buffer_rsrc_t dgR = __builtin_amdgcn_make_buffer_rsrc(
        p.dgates, 0, (int)((u32)T * fdg), 0x00020000);
    auto ldf = [&](__amdgpu_buffer_rsrc_t R, u32 voff, u32 soff) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, voff, soff, 0));
    };
    struct Pre { u32x2 g[NE]; float cp[NE], dy[NE], dyb[NE]; };     // gates stay packed until used
    auto fetch = [&](int t, Pre &q) {
        const int tp = dir == 0 ? t - 1 : t + 1;
        const int tpc = tp < 0 ? 0 : (tp >= T ? T - 1 : tp);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            q.g[e] = __builtin_amdgcn_raw_buffer_load_b64(gR, vcs[e] * 2u, (u32)t * fg, 0);
            q.cp[e] = ldf(csR, vcs[e], (u32)tpc * fcs);
            q.dy[e] = ldf(dyR, vdy[e], (u32)t * fdy);
            q.dyb[e] = ldf(dyR, vdy[e], (u32)t * fdy + dy2);
        }
    };
    Pre cur;
    float pcs[NE];
    {
        const int t0 = dir == 0 ? T - 1 : 0;
        fetch(t0, cur);
#pragma unroll
        for (int e = 0; e < NE; ++e) pcs[e] = ldf(csR, vcs[e], (u32)t0 * fcs);
    }
    __bf16 sod[NE][4] = {};
    int st = 0;
    auto bulk_store = [&](bool live) {      // live == false: offsets out of range (first step)
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b16(
                    (short)__builtin_bit_cast(unsigned short, sod[e][g]), dgR, live ? vdg[e] : OOBV,
                    (u32)st * fdg + (u32)g * H * 2u, 0);
    };

    PSTAMP_DECL;
    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;
        if (step > 0 && tid == 0 && !dead_s) {
            if (!team_wait(myctr, (unsigned)(njt * step), ctl.spin_limit, ctl.err)) dead_s = 1;
        }
        PSTAMP(0);
        __syncthreads();
        // ---- dgates_{prev step} rows of this batch tile, all 4H columns (4*KS KiB): the
        // 16 KiB this workgroup produced itself are still in dg_lds (LDS -> LDS copy), the
        // rest comes from the team mates over sc1 loads
        // (real batch rows only, as in the forward kernel)
        constexpr int RL = 8 * NE, CPK = 2 * RL;
        constexpr int CHO = (KS4 - 16) * CPK, NI = (CHO + 511) / 512;
        u32x4 tmp[NI > 0 ? NI : 1];
        if constexpr (NI > 0) {
            const unsigned base = (unsigned)((((size_t)(step & 1) * 2 + dir) * Bp * H4 +
                                              (size_t)btile * KS4 * 512) * 2);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int co = i * 512 + tid;
                const int cc = co < CHO ? co : CHO - 1;
                const int kq = cc / CPK, q = cc - kq * CPK, g = kq / (KS - 4), r = kq - g * (KS - 4);
                const int k = g * KS + (r < 4 * jt ? r : r + 4);
                tmp[i] = __builtin_amdgcn_raw_buffer_load_b128(
                    dres, base + (k * 64 + (q < RL ? q : q + (32 - RL))) * 16, 0, ASR_SC1);
            }
        }
        bulk_store(step > 0);     // previous step's dgates rows, under the tile-load latency
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 512 + tid, bi = c >> 6;         // bi = gate * 4 + kk
            reinterpret_cast<u32x4 *>(a_lds)[((bi >> 2) * KS + 4 * jt + (bi & 3)) * 64 + (c & 63)] =
                reinterpret_cast<const u32x4 *>(dg_lds)[c];
        }
        if constexpr (NI > 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int co = i * 512 + tid;
                const int kq = co / CPK, q = co - kq * CPK, g = kq / (KS - 4), r = kq - g * (KS - 4);
                const int k = g * KS + (r < 4 * jt ? r : r + 4);
                if (co < CHO)
                    reinterpret_cast<u32x4 *>(a_lds)[k * 64 + (q < RL ? q : q + (32 - RL))] = tmp[i];
            }
        }
        __syncthreads();
        PSTAMP(1);
        Pre nxt;
        {
            const int sn = step + 1 < T ? step + 1 : step;
            fetch(dir == 0 ? T - 1 - sn : sn, nxt);
        }
        {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + (size_t)kq * KS * 64 + lane;
            acc = mfma_chain<0, KS, 4>(al, fb, acc);
            const int c32 = lane & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                part[kq][row][js * 32 + c32] = acc[i];
            }
        }
        __syncthreads();
        PSTAMP(2);
        const bool dead = dead_s != 0;
        st = t;
        const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = e * 8 + wave;
            float od[4];
            if (t < len[e]) {
                const float dh = (cur.dy[e] + cur.dyb[e]) + (part[0][row][col] + part[1][row][col]) +
                                 (part[2][row][col] + part[3][row][col]);
                const float cp = (tp >= 0 && tp < len[e]) ? cur.cp[e] : 0.f;
                float dcout, gt[4];
                unpack_gates(cur.g[e], gt);
                lstm_cell_bwd(gt, pcs[e], cp, dh, dcarry[e], od, dcout);
                dcarry[e] = dcout;
                if (dead) od[0] = od[1] = od[2] = od[3] = __builtin_nanf("");
            } else {
                // no gradient reaches a padding frame; the carried state gradient restarts from zero
                od[0] = od[1] = od[2] = od[3] = 0.f;
                dcarry[e] = 0.f;
            }
            const int slot = ((col >> 4) * 64 + row + 32 * ((col >> 3) & 1)) * 8 + (col & 7);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sod[e][g] = (__bf16)od[g];
                dg_lds[g * 2048 + slot] = sod[e][g];
            }
            pcs[e] = cur.cp[e];         // c of the frame the next step visits
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int bi = wave * 2 + r, g = bi >> 2, kk = bi & 3;
            const u32x4 v = reinterpret_cast<const u32x4 *>(dg_lds)[bi * 64 + lane];
            const unsigned off = (unsigned)(((((size_t)((step + 1) & 1) * 2 + dir) * Bp * H4) +
                                             ((size_t)btile * KS4 + g * KS + 4 * jt + kk) * 512 + lane * 8) * 2);
            __builtin_amdgcn_raw_buffer_store_b128(v, dres, (lane & 31) < 8 * NE ? off : 0xFFFFFFFFu,
                                                   0, ASR_SC1);
        }
        PSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PSTAMP(4);
        if (tid == 0) __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = nxt;
        PSTAMP(5);
    }
    if (T > 0) bulk_store(true);          // the last step's
#ifdef ASR_LSTM_STAMPS
    __syncthreads();
    if (tid == 0 && b0 < B)
        for (int i = 0; i < 6; ++i)
            p.dgates[(((size_t)0 * B + b0) * 2 + dir) * H4 + j0 + i] = (__bf16)((float)pst[i] / T / 16.f);
#endif
}

// Backward recurrence WITH the input gradient: dx_t = dgates_t · W_ih per direction
// (replaces the [T·B, 8H] x [8H, F] library GEMM behind asr_lstm_bidir_bwd_bf16 when F == H).
// MEASURED SLOWER than recurrence + GEMM at the bench shape (B=576, H=320: 1669 us against
// 1246 + 331): unlike the forward kernel, whose hand-off waits were idle, every phase of the
// backward step is bound by what a wave has to ISSUE, so the dx product (+800 cycles per
// step beyond the team wait it overlaps), the partial-sum read-out (+500) and the operand
// DMAs (+700) all add to the step (ablation stamps).  Callers opt in
// (att_speech: ASR_LSTM_FUSED_BWD=1); the parity tests run it either way.
// Same team protocol, tiles and arithmetic as lstm_bwd_persist_kernel; what differs:
//  * the team's dgates tile of step s stays in a_lds until the tile of step s+1 replaces it,
//    so every wave multiplies it with its W_ihᵀ slice (registers, next to the W_hhᵀ slice) at
//    the top of step s+2 — in the window in which the workgroup waits for the team counter
//    anyway; the four K-quarter partials meet in `part` (free between the cell phase and the
//    next dh product) and leave as one coalesced fp32 row per wave.  The two directions write
//    separate planes; the layer below reads both (dy_shared == 2);
//  * two weight slices take 160 of the 256 VGPRs, so nothing else may be staged in registers:
//    the hand-off tile comes global -> LDS by LDS-DMA (the buffer is fragment-major, 1 KiB
//    per k-step, lane-linear on both sides), and so do the pointwise operands of the next
//    step (saved gates, neighbour cell state, dy): issued behind the team signal, landed
//    under the dx product, read from LDS in the cell phase;
//  * every per-row quantity (lengths, row offsets) is wave-uniform and lives in SGPRs.
template <int KS, int NE>
__global__ __launch_bounds__(512) void lstm_bwd_dx_kernel(LstmBwdParams p, LstmTeamCtl ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS4 = 4 * KS, RL = 8 * NE;
    __bf16 *a_lds = reinterpret_cast<__bf16 *>(smem);                       // 4*KS KiB
    float (*part)[32][65] = reinterpret_cast<float (*)[32][65]>(smem + KS4 * 1024);
    __bf16 *dg_lds = reinterpret_cast<__bf16 *>(smem + KS4 * 1024 + ASR_GLDS_BYTES);   // 16 KiB
    unsigned char *pre = smem + KS4 * 1024 + ASR_GLDS_BYTES + 16384;
    u32x2 (*pg)[64] = reinterpret_cast<u32x2 (*)[64]>(pre);                 // [RL][64] gate records
    float (*pcp)[64] = reinterpret_cast<float (*)[64]>(pre + RL * 512);     // [RL][64] neighbour c
    float (*pdy)[64] = reinterpret_cast<float (*)[64]>(pre + RL * 768);     // [2][RL][64] dy planes
    __shared__ int dead_s;
    typedef unsigned int u32;
    constexpr u32 OOBV = 0x80000000u;
    const int H = p.H, B = p.B, T = p.T, H4 = 4 * p.H;
    int jt = blockIdx.x, btile = blockIdx.y + ctl.bt0, dir = blockIdx.z;
    if (ctl.xcd_teams) {                     // XCD-affine 1-D grid, as in lstm_bwd_persist_kernel
        const int slot = blockIdx.x >> 3, team = (slot / (p.H / 64)) * 8 + (blockIdx.x & 7);
        if (team >= ctl.xcd_teams) return;
        jt = slot % (p.H / 64);
        btile = (team >> 1) + ctl.bt0;
        dir = team & 1;
    }
    const int j0 = jt * 64, b0 = btile * RL, njt = H / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave >> 1, js = wave & 1;
    const size_t Bp = ctl.rows;
    unsigned *myctr = ctl.ctr + ((size_t)dir * ctl.nbt + btile) * 32;
    if (tid == 0) dead_s = 0;
    for (int i = tid; i < 8192; i += 512) dg_lds[i] = (__bf16)0.f;      // padding rows stay 0
    for (int i = tid; i < KS4 * 64; i += 512)
        reinterpret_cast<u32x4 *>(a_lds)[i] = u32x4{0u, 0u, 0u, 0u};

    bf16x8 fb[KS], fxb[KS];
    {
        const size_t wo = (size_t)dir * H * H4 +
                          (((size_t)(2 * jt + js) * KS4 + (size_t)kq * KS) * 64 + lane) * 8;
#pragma unroll
        for (int k = 0; k < KS; ++k) fb[k] = *reinterpret_cast<const bf16x8 *>(p.whhT + wo + k * 512);
#pragma unroll
        for (int k = 0; k < KS; ++k) fxb[k] = *reinterpret_cast<const bf16x8 *>(p.wihT + wo + k * 512);
    }
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(
        p.dgbuf, 0, (int)(2 * 2 * Bp * H4 * 2), 0x00020000);
    const rsrc_words dresD = raw_rsrc(p.dgbuf, (unsigned)(2 * 2 * Bp * H4 * 2));

    // rows of this wave: b = b0 + e*8 + wave (wave-uniform: SGPRs)
    const int col = lane;
    int len[NE], bc[NE];
    bool inb[NE];
    float dcarry[NE], pcs[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int b = b0 + e * 8 + wave;
        inb[e] = b < B;
        bc[e] = b < B ? b : B - 1;
        len[e] = __builtin_amdgcn_readfirstlane(b < B ? p.lens[bc[e]] : 0);
        dcarry[e] = 0.f;
    }
    const u32 l4 = (u32)lane * 4u, l2 = (u32)lane * 2u;
    const u32 fg = (u32)B * 2u * H * 8u, fcs = (u32)B * 2u * H * 4u;
    const u32 fdy = (u32)B * H * 4u * (p.dy_shared ? 1u : 2u), fdg = (u32)B * 2u * H4 * 2u;
    const u32 fdx = (u32)B * H * 4u;
    const rsrc_words gD = raw_rsrc(p.gates, (u32)T * fg);
    const rsrc_words csD = raw_rsrc(p.csave, (u32)T * fcs);
    const rsrc_words dyD = raw_rsrc(p.dy, (u32)T * fdy * (p.dy_shared == 2 ? 2u : 1u));
    const __amdgpu_buffer_rsrc_t dgR = __builtin_amdgcn_make_buffer_rsrc(
        p.dgates, 0, (int)((u32)T * fdg), 0x00020000);
    const __amdgpu_buffer_rsrc_t dxR = __builtin_amdgcn_make_buffer_rsrc(
        p.dx, 0, (int)(2u * (u32)T * fdx), 0x00020000);

    // pointwise operands of frame t (all RL rows of the tile) -> LDS: 5*RL/4 LDS-DMAs of
    // 1 KiB (two rows of gate records, or four rows of c / dy), dealt round-robin to the
    // waves; the per-lane source offsets are loop-invariant
    constexpr int NQ = 5 * RL / 4, NQW = (NQ + 7) / 8;
    u32 pvo[NQW];
#pragma unroll
    for (int i = 0; i < NQW; ++i) {
        const int q = wave + 8 * i;
        if (q < RL / 2) {
            const int r = 2 * q + (lane >> 5), b = b0 + r < B ? b0 + r : B - 1;
            pvo[i] = (u32)((dir * B + b) * H + j0) * 8u + (u32)(lane & 31) * 16u;
        } else {
            const int qq = q < 3 * RL / 4 ? q - RL / 2 : (q < RL ? q - 3 * RL / 4 : q - RL);
            const int r = 4 * qq + (lane >> 4), b = b0 + r < B ? b0 + r : B - 1;
            if (q < 3 * RL / 4) pvo[i] = (u32)((dir * B + b) * H + j0) * 4u + (u32)(lane & 15) * 16u;
            else pvo[i] = (u32)(p.dy_shared ? b * H + j0 : (b * 2 + dir) * H + j0) * 4u + (u32)(lane & 15) * 16u;
            if (q >= RL && p.dy_shared != 2) pvo[i] = OOBV;      // no second dy plane: zeros
        }
    }
    auto pre_dma = [&](int t) {
        const int tp = dir == 0 ? t - 1 : t + 1;
        const int tpc = tp < 0 ? 0 : (tp >= T ? T - 1 : tp);
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            const int q = wave + 8 * i;
            if (q < RL / 2) dma16_s(gD, lds_addr(pre) + (unsigned)q * 1024u, pvo[i], (u32)t * fg);
            else if (q < 3 * RL / 4) dma16_s(csD, lds_addr(pre) + (unsigned)q * 1024u, pvo[i], (u32)tpc * fcs);
            else if (q < RL) dma16_s(dyD, lds_addr(pre) + (unsigned)q * 1024u, pvo[i], (u32)t * fdy);
            else if (q < NQ) dma16_s(dyD, lds_addr(pre) + (unsigned)q * 1024u, pvo[i], (u32)(T + t) * fdy);
        }
    };
    __bf16 sod[NE][4] = {};
    int st = 0;
    auto bulk_store = [&](bool live) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b16(
                    (short)__builtin_bit_cast(unsigned short, sod[e][g]), dgR,
                    live && inb[e] ? l2 : OOBV,
                    (u32)st * fdg + (u32)(((b0 + e * 8 + wave) * 2 + dir) * H4 + g * H + j0) * 2u, 0);
    };
    // a_lds (the dgates of one step, all 4H columns) x this wave's W_ihᵀ slice -> part
    auto dx_mfma = [&]() {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + (size_t)kq * KS * 64 + lane;
        acc = mfma_chain<0, KS, 2>(al, fxb, acc);
        const int c32 = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            part[kq][row][js * 32 + c32] = acc[i];
        }
    };
    // the four K-quarter partials in `part` -> dx[dir][frame of step sq]
    auto dx_store = [&](bool live, int sq) {
        const int tq = live ? (dir == 0 ? T - 1 - sq : sq) : 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = e * 8 + wave;
            const float v = (part[0][row][col] + part[1][row][col]) + (part[2][row][col] + part[3][row][col]);
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(int, v), dxR, live && inb[e] ? l4 : OOBV,
                (u32)(dir * T + tq) * fdx + (u32)((b0 + e * 8 + wave) * H + j0) * 4u, 0);
        }
    };
    // the other workgroups' 4*(KS-4) k-steps of the team's dgates tile, (KS-4)/2 per wave
    auto tile_dma = [&](int step) {
        const unsigned base = (unsigned)((((size_t)(step & 1) * 2 + dir) * Bp * H4 +
                                          (size_t)btile * KS4 * 512) * 2);
        const u32 voff = (lane & 31) < RL ? (u32)lane * 16u : OOBV;
        if constexpr (KS > 4) {
            // a wave's k-steps come in runs of four consecutive ones when (KS-4)/2 % 4 == 0
            // (the gap left by this workgroup's own k-steps starts at a multiple of four)
            constexpr int PW = (KS - 4) / 2, GR = PW % 4 == 0 ? 4 : 1;
#pragma unroll
            for (int i = 0; i < PW; i += GR) {
                const int ko = wave * PW + i, g = ko / (KS - 4), r = ko - g * (KS - 4);
                const int k = g * KS + (r < 4 * jt ? r : r + 4);
                if constexpr (GR == 4)
                    dma16x4_sc1(dresD, lds_addr(a_lds) + (unsigned)k * 1024u, voff, base + (unsigned)k * 1024u);
                else
                    dma16_sc1(dresD, lds_addr(a_lds) + (unsigned)k * 1024u, voff, base + (unsigned)k * 1024u);
            }
        }
    };
    auto own_copy = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 512 + tid, bi = c >> 6;         // bi = gate * 4 + kk
            reinterpret_cast<u32x4 *>(a_lds)[((bi >> 2) * KS + 4 * jt + (bi & 3)) * 64 + (c & 63)] =
                reinterpret_cast<const u32x4 *>(dg_lds)[c];
        }
    };
    {
        const int t0 = dir == 0 ? T - 1 : 0;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            pcs[e] = p.csave[(((size_t)t0 * 2 + dir) * B + bc[e]) * H + j0 + col];
    }
    const bool last_q = wave + 8 * (NQW - 1) < NQ;      // does this wave issue NQW operand DMAs, or one less
    // everything loaded so far lands before the loop: the compiler cannot count the LDS-DMA
    // inside the asm (see lstm_fwd_persist_kernel)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        asm volatile("" : "+v"(fb[k]));
        asm volatile("" : "+v"(fxb[k]));
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) asm volatile("" : "+v"(pcs[e]));

    PSTAMP_DECL;
    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;
        // The team counter is looked at before and in the middle of the dx product (a_lds still
        // holds the dgates of step - 2, loaded for the dh product of step - 1): the round trip
        // of the poll runs under the MFMAs instead of after them
        const unsigned target = (unsigned)(njt * step);
        const bool poller = step > 0 && tid == 0 && !dead_s;
        unsigned seen = 0;
        if (poller) seen = __hip_atomic_load(myctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (step >= 2) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + (size_t)kq * KS * 64 + lane;
            acc = mfma_chain<0, KS / 2, 2>(al, fxb, acc);
            if (poller && seen < target)
                seen = __hip_atomic_load(myctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = mfma_chain<KS / 2, KS, 2>(al, fxb, acc);
            const int c32 = lane & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                part[kq][row][js * 32 + c32] = acc[i];
            }
        }
        if (poller && seen < target) {
            if (!team_wait(myctr, target, ctl.spin_limit, ctl.err)) dead_s = 1;
        }
        PSTAMP(0);
        __syncthreads();
        tile_dma(step);
        bulk_store(step > 0);     // previous step's dgates rows, under the tile's flight
        own_copy();
        dx_store(step >= 2, step - 2);
        pre_dma(t);               // this step's pointwise operands: needed two barriers further on
        // the tile has landed once all but the stores and operand DMAs behind it are done
        // (VMEM retires in issue order)
        if (last_q) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NE + NQW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NE + NQW - 1) : "memory");
        __syncthreads();
        PSTAMP(1);
        {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const bf16x8 *al = reinterpret_cast<const bf16x8 *>(a_lds) + (size_t)kq * KS * 64 + lane;
            acc = mfma_chain<0, KS, 2>(al, fb, acc);
            const int c32 = lane & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                part[kq][row][js * 32 + c32] = acc[i];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the operand DMAs
        __syncthreads();
        PSTAMP(2);
        const bool dead = dead_s != 0;
        st = t;
        const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = e * 8 + wave;
            float od[4];
            const float cpn = pcp[row][col];
            if (t < len[e]) {
                const float dh = (pdy[row][col] + pdy[RL + row][col]) +
                                 (part[0][row][col] + part[1][row][col]) +
                                 (part[2][row][col] + part[3][row][col]);
                const float cp = (tp >= 0 && tp < len[e]) ? cpn : 0.f;
                float dcout, gt[4];
                unpack_gates(pg[row][col], gt);
                lstm_cell_bwd(gt, pcs[e], cp, dh, dcarry[e], od, dcout);
                dcarry[e] = dcout;
                if (dead) od[0] = od[1] = od[2] = od[3] = __builtin_nanf("");
            } else {
                od[0] = od[1] = od[2] = od[3] = 0.f;
                dcarry[e] = 0.f;
            }
            const int slot = ((col >> 4) * 64 + row + 32 * ((col >> 3) & 1)) * 8 + (col & 7);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sod[e][g] = (__bf16)od[g];
                dg_lds[g * 2048 + slot] = sod[e][g];
            }
            pcs[e] = cpn;               // c of the frame the next step visits
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int bi = wave * 2 + r, g = bi >> 2, kk = bi & 3;
            const u32x4 v = reinterpret_cast<const u32x4 *>(dg_lds)[bi * 64 + lane];
            const unsigned off = (unsigned)(((((size_t)((step + 1) & 1) * 2 + dir) * Bp * H4) +
                                             ((size_t)btile * KS4 + g * KS + 4 * jt + kk) * 512 + lane * 8) * 2);
            __builtin_amdgcn_raw_buffer_store_b128(v, dres, (lane & 31) < RL ? off : 0xFFFFFFFFu,
                                                   0, ASR_SC1);
        }
        PSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PSTAMP(4);
        if (tid == 0) __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PSTAMP(5);
    }
    bulk_store(true);                     // the last step's dgates rows
    // dx of the last two steps: step T-2's tile is in a_lds, step T-1's comes with one more
    // hand-off round
    if (T >= 2) dx_mfma();
    if (tid == 0 && !dead_s) {
        if (!team_wait(myctr, (unsigned)(njt * T), ctl.spin_limit, ctl.err)) dead_s = 1;
    }
    __syncthreads();
    tile_dma(T);
    own_copy();
    dx_store(T >= 2, T - 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NE) : "memory");
    __syncthreads();
    dx_mfma();
    __syncthreads();
    dx_store(true, T - 1);
#ifdef ASR_LSTM_STAMPS
    __syncthreads();
    if (tid == 0 && b0 < B)
        for (int i = 0; i < 6; ++i)
            p.dgates[(((size_t)0 * B + b0) * 2 + dir) * H4 + j0 + i] = (__bf16)((float)pst[i] / T / 16.f);
#endif
}

// several ranges in ONE launch (a forward call clears its state buffers, the four pad frames
// of y_bf16 and the team counters: six 5-us launches per layer and direction pair otherwise)
struct ZeroRanges { uint32_t *p[6]; size_t n[6]; int count; };
__global__ void zero_ranges_kernel(ZeroRanges z) {
    for (int r = 0; r < z.count; ++r) {
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (; i < z.n[r]; i += stride) z.p[r][i] = 0u;
    }
}
struct ZeroList {
    ZeroRanges z;
    ZeroList() { z.count = 0; }
    void add(void *p, size_t bytes) {
        if (bytes / 4 == 0) return;
        z.p[z.count] = (uint32_t *)p; z.n[z.count] = bytes / 4; ++z.count;
    }
    void launch(hipStream_t s) {
        if (!z.count) return;
        size_t most = 0;
        for (int r = 0; r < z.count; ++r) most = z.n[r] > most ? z.n[r] : most;
        int blocks = (int)((most + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(zero_ranges_kernel, dim3(blocks), dim3(256), 0, s, z);
    }
};

inline int64_t ctl_bytes(int B) { return ((int64_t)2 * ((B + 15) / 16) * 128 + 256 + 255) / 256 * 256; }

// persistent path on unless ASR_LSTM_PERSIST=0 (A/B switch for tests and profiling)
inline bool persist_enabled() {
    const char *e = getenv("ASR_LSTM_PERSIST");
    return !(e && e[0] == '0');
}

inline int cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

// Launch a persistent kernel over all batch tiles, at most one workgroup per CU
// per launch (every team of a launch must be resident).  Returns false if the
// shape cannot run persistently (caller falls back to one launch per step).
// rows of one hbuf / dgbuf plane: 32 per batch tile of the smallest tile (16 rows), or
// the 64-row padding of the per-step kernels, whichever is larger
inline int64_t plane_rows(int B) {
    const int64_t a = (int64_t)(B + 63) / 64 * 64, b = (int64_t)((B + 15) / 16) * 32;
    return a > b ? a : b;
}

inline int pick_tile_rows(int B, int njt, int cus) {
    for (int cand = 2; cand <= 4; ++cand)
        if (2 * njt * ((B + 8 * cand - 1) / (8 * cand)) <= cus) return cand;
    return 4;
}

template <typename P>
bool launch_persist(void (*const kerns[3])(P, LstmTeamCtl), const P &p, int B, int H,
                    size_t lds_need, unsigned *ctl_words, unsigned *err_flag, hipStream_t s) {
    const int njt = H / 64;
    const int cus = cu_count();
    if ((H % 64) != 0 || njt < 1 || cus < 2 * njt || lds_need > 160 * 1024) return false;
    // batch-tile rows: the smallest of 16 / 24 / 32 whose whole grid is one launch with one
    // workgroup per CU (more workgroups = less work on each one's critical path); 32-row
    // tiles in several launches when the batch is too large for that
    const int ne = pick_tile_rows(B, njt, cus);
    void (*kern)(P, LstmTeamCtl) = kerns[ne - 2];
    if (!kern) return false;
    const int nbt = (B + 8 * ne - 1) / (8 * ne);
    const size_t lds = lds_need > 84 * 1024 ? lds_need : 84 * 1024;     // > half the LDS: 1 workgroup per CU
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return false;
    const int max_bt = cus / (2 * njt);
    LstmTeamCtl ctl;
    ctl.ctr = ctl_words + 64;
    ctl.err = err_flag ? err_flag : ctl_words;
    ctl.spin_limit = 1u << 18;
    if (const char *e = getenv("ASR_LSTM_SPIN_LIMIT")) ctl.spin_limit = (unsigned)strtoul(e, nullptr, 10);
    ctl.nbt = nbt;
    ctl.rows = (size_t)plane_rows(B);
    for (int bt0 = 0; bt0 < nbt; bt0 += max_bt) {
        ctl.bt0 = bt0;
        const int n = nbt - bt0 < max_bt ? nbt - bt0 : max_bt;
        // XCD-affine placement of the teams when the grid padded to a multiple of 8 teams
        // still is one workgroup per CU (ASR_LSTM_XCD=0: plain 3-D grid)
        const int padded = 8 * njt * ((2 * n + 7) / 8);
        static const bool xcd_on = !(getenv("ASR_LSTM_XCD") && getenv("ASR_LSTM_XCD")[0] == '0');
        if (xcd_on && padded <= cus) {
            ctl.xcd_teams = 2 * n;
            hipLaunchKernelGGL(kern, dim3(padded), dim3(512), lds, s, p, ctl);
        } else {
            ctl.xcd_teams = 0;
            hipLaunchKernelGGL(kern, dim3(njt, n, 2), dim3(512), lds, s, p, ctl);
        }
    }
    return true;
}

}  // namespace

extern "C" int64_t asr_lstm_workspace_bytes(int B, int H) {
    if (B < 0 || H < 0) return -1;
    // forward: hbuf bf16 [2][2][Bp][H] + cbuf f32 [2][B][H] + packed W_hh
    // backward: dgbuf bf16 [2][2][Bp][4H] + dcbuf f32 [2][B][H] + packed W_hhᵀ (larger)
    // + team counters of the persistent kernels: [2 dir][ceil(B/32)] x 128 B and a timeout word
    const int64_t Bp = plane_rows(B);
    // (forward with the fused input projection: a second packed matrix, W_ih)
    return (int64_t)2 * 2 * Bp * 4 * H * 2 + (int64_t)2 * B * H * 4 +
           (int64_t)2 * 4 * H * H * 2 + (int64_t)2 * 4 * H * (H + 64) * 2 + 256 + ctl_bytes(B);
}

namespace {
int lstm_fwd_impl(const void *gx, int gx_bf16, const void *x_bf16, const void *wih_bf16, int F,
                  const void *whh_bf16, const int32_t *lens, int T, int B, int H,
                  float *y, void *y_bf16, void *gates_bf16, float *csave, void *xsum_bf16,
                  void *workspace, int64_t workspace_bytes, uint32_t *err_flag, void *stream);
}

extern "C" int asr_lstm_bidir_fwd_bf16(const void *gx, int gx_bf16, const void *whh_bf16,
                                       const int32_t *lens, int T, int B, int H,
                                       float *y, void *y_bf16, void *gates_bf16,
                                       float *csave,
                                       void *workspace, int64_t workspace_bytes,
                                       uint32_t *err_flag, void *stream) {
    if (!gx) return ASR_EINVAL;
    return lstm_fwd_impl(gx, gx_bf16, nullptr, nullptr, 0, whh_bf16, lens, T, B, H, y, y_bf16,
                         gates_bf16, csave, nullptr, workspace, workspace_bytes, err_flag, stream);
}

extern "C" int asr_lstm_bidir_fwd_fused_bf16(const void *x_bf16, const void *wih_bf16,
                                             const void *whh_bf16, const int32_t *lens,
                                             int T, int B, int H, int F, float *y, void *y_bf16,
                                             void *gates_bf16, float *csave, void *workspace,
                                             int64_t workspace_bytes, uint32_t *err_flag,
                                             void *stream) {
    if (!x_bf16 || !wih_bf16) return ASR_EINVAL;
    return lstm_fwd_impl(nullptr, 1, x_bf16, wih_bf16, F, whh_bf16, lens, T, B, H, y, y_bf16,
                         gates_bf16, csave, nullptr, workspace, workspace_bytes, err_flag, stream);
}

#ifdef ASR_EXPERIMENTS   // include/asr_amd_experiments.h: not in the default library
// ... and xsum [T,B,H] bf16 = bf16(h_fwd) + bf16(h_rev), the next layer's input (BatchRNN's
// direction merge, encoder_utils.py:112-117, on the bf16 planes): the recurrence runs as TWO
// launches, steps [0, ceil(T/2)) and the rest; in the second every frame a direction reaches
// was written by the other one in the first, so it adds that value to its own output — no
// separate pass over the planes (0.1 ms per layer at B=768), no cross-direction hand-shake.
// For odd T the middle frame is reached by both directions in the first launch: it is left
// to the caller (xsum[T/2] is not written).
extern "C" int asr_lstm_bidir_fwd_fused_sum_bf16(const void *x_bf16, const void *wih_bf16,
                                                 const void *whh_bf16, const int32_t *lens,
                                                 int T, int B, int H, int F, float *y, void *y_bf16,
                                                 void *gates_bf16, float *csave, void *xsum_bf16,
                                                 void *workspace, int64_t workspace_bytes,
                                                 uint32_t *err_flag, void *stream) {
    if (!x_bf16 || !wih_bf16 || !xsum_bf16) return ASR_EINVAL;
    if (T < 2) return ASR_EUNSUPPORTED;
    return lstm_fwd_impl(nullptr, 1, x_bf16, wih_bf16, F, whh_bf16, lens, T, B, H, y, y_bf16,
                         gates_bf16, csave, xsum_bf16, workspace, workspace_bytes, err_flag, stream);
}
#endif

extern "C" int asr_lstm_fused_supported(int B, int H, int F) {
    if (!persist_enabled() || B <= 0) return 0;
    if (H != 64 && H != 128 && H != 256 && H != 320) return 0;
    // input size: the hidden size, or (H = 320) the 352 features of the conv front-end —
    // forward only
    // bit 2: asr_lstm_bidir_fwd_fused_sum_bf16 (H = 320; not the 352-feature layer at 32-row tiles)
    if (F != H)
        return (H == 320 && F == 352 && cu_count() >= 2 * (H / 64))
                   ? 1 | (pick_tile_rows(B, H / 64, cu_count()) < 4 ? 4 : 0) : 0;
    const int cus = cu_count(), njt = H / 64;
    if (cus < 2 * njt) return 0;
    // bit 0: forward (input projection); bit 1: backward (input gradient) — its kernel has no
    // LDS left for 32-row batch tiles, i.e. the batch must fit one launch of 16/24-row tiles
    return 1 | (pick_tile_rows(B, njt, cus) < 4 ? 2 : 0) | (H == 320 ? 4 : 0);
}

namespace {
int lstm_fwd_impl(const void *gx, int gx_bf16, const void *x_bf16, const void *wih_bf16, int F,
                  const void *whh_bf16, const int32_t *lens, int T, int B, int H,
                  float *y, void *y_bf16, void *gates_bf16, float *csave, void *xsum_bf16,
                  void *workspace, int64_t workspace_bytes, uint32_t *err_flag, void *stream) {
    const bool fused = x_bf16 != nullptr;
    if (xsum_bf16 && !fused) return ASR_EINVAL;
    if (T < 0 || B <= 0 || H <= 0 || (H % 32) != 0) return ASR_EINVAL;
    if (T == 0) return ASR_OK;
    if (!whh_bf16 || !lens || !y_bf16 || !gates_bf16 || !csave || !workspace)
        return ASR_EINVAL;
    if (fused && (!(asr_lstm_fused_supported(B, H, F) & 1) || (uint64_t)T * B * F * 2 >= (1ull << 31)))
        return ASR_EUNSUPPORTED;
    if (fused && F != H && y) return ASR_EINVAL;      // that variant writes the bf16 outputs only
    if (workspace_bytes < asr_lstm_workspace_bytes(B, H)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    LstmFwdParams p;
    const size_t Bp = (size_t)plane_rows(B);
    const size_t hbytes = (size_t)2 * 2 * Bp * H * 2, cbytes = (size_t)2 * B * H * 4;
    __bf16 *wpack = (__bf16 *)((char *)workspace + hbytes + cbytes);
    p.gx = gx; p.gx_bf16 = gx_bf16; p.whh = wpack; p.lens = lens;
    p.x = (const __bf16 *)x_bf16; p.wih = wpack + (size_t)8 * H * H;
    p.T = T; p.B = B; p.H = H;
    p.hbuf = (__bf16 *)workspace;
    p.cbuf = (float *)((char *)workspace + hbytes);
    p.y = y; p.ybf = (__bf16 *)y_bf16; p.gates = (u32x2 *)gates_bf16; p.csave = csave;
    p.step = 0; p.s_begin = 0; p.s_end = T; p.xsum = nullptr;
    ZeroList zl;
    zl.add(workspace, hbytes + cbytes);
    // the two pad frames of every direction of y_bf16
    for (int d = 0; d < 2; ++d) {
        zl.add(p.ybf + (size_t)d * (T + 2) * B * H, (size_t)B * H * 2);
        zl.add(p.ybf + ((size_t)d * (T + 2) + T + 1) * B * H, (size_t)B * H * 2);
    }
    if (persist_enabled())
        zl.add((char *)workspace + asr_lstm_workspace_bytes(B, H) - ctl_bytes(B), (size_t)ctl_bytes(B));
    zl.launch(s);
    // [2 dir x 4 gates] matrices of H x H (rows = hidden unit, cols = k)
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, s,
                       (const __bf16 *)whh_bf16, wpack, 8, H, H, 0);
    if (fused)          // wih_bf16: [2 dir][4H rows (gate-major)][F cols] row-major
        hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, s,
                           (const __bf16 *)wih_bf16, wpack + (size_t)8 * H * H, 8, H, F, 0);
    if (persist_enabled()) {
        unsigned *ctl_words = (unsigned *)((char *)workspace + asr_lstm_workspace_bytes(B, H) - ctl_bytes(B));
        void (*pk[3])(LstmFwdParams, LstmTeamCtl) = {nullptr, nullptr, nullptr};
#define ASR_PICK(KSV) if (H == 16 * KSV && !fused) {                                           \
        if (gx_bf16) { pk[0] = lstm_fwd_persist_kernel<KSV, 2, 1, 0>; pk[1] = lstm_fwd_persist_kernel<KSV, 3, 1, 0>; \
                       pk[2] = lstm_fwd_persist_kernel<KSV, 4, 1, 0>; }                             \
        else { pk[0] = lstm_fwd_persist_kernel<KSV, 2, 0, 0>; pk[1] = lstm_fwd_persist_kernel<KSV, 3, 0, 0>; \
               pk[2] = lstm_fwd_persist_kernel<KSV, 4, 0, 0>; } }
        ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16) ASR_PICK(24) ASR_PICK(32)   // 48: W_hh slice spills
#undef ASR_PICK
#define ASR_PICK(KSV) if (H == 16 * KSV && fused && F == H) { pk[0] = lstm_fwd_persist_kernel<KSV, 2, 1, KSV>; \
        pk[1] = lstm_fwd_persist_kernel<KSV, 3, 1, KSV>; pk[2] = lstm_fwd_persist_kernel<KSV, 4, 1, KSV>; }
        ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16)     // two weight slices in registers
#undef ASR_PICK
        if (H == 320 && fused && F == 352) {                  // the first layer behind the conv front-end
            pk[0] = lstm_fwd_persist_kernel<20, 2, 1, 22>; pk[1] = lstm_fwd_persist_kernel<20, 3, 1, 22>;
            pk[2] = lstm_fwd_persist_kernel<20, 4, 1, 22>;
        }
        // the second launch of the direction-sum variant (H = 320 only; the 32-row tile kernel
        // of the 352-feature layer has no register left for it: that layer's caller adds)
        void (*pks[3])(LstmFwdParams, LstmTeamCtl) = {nullptr, nullptr, nullptr};
        if (H == 320 && fused && F == H) {
            pks[0] = lstm_fwd_persist_kernel<20, 2, 1, 20, 1>; pks[1] = lstm_fwd_persist_kernel<20, 3, 1, 20, 1>;
            pks[2] = lstm_fwd_persist_kernel<20, 4, 1, 20, 1>;
        }
        if (H == 320 && fused && F == 352) {
            pks[0] = lstm_fwd_persist_kernel<20, 2, 1, 22, 1>; pks[1] = lstm_fwd_persist_kernel<20, 3, 1, 22, 1>;
        }
        if (xsum_bf16 && !pks[pick_tile_rows(B, H / 64, cu_count()) - 2]) return ASR_EUNSUPPORTED;
        // the persistent kernel addresses gx / y / gates / ... with 32-bit byte offsets
        const bool fits32 = (uint64_t)T * B * 8 * H * 4 < (1ull << 32) &&
                            (uint64_t)2 * (T + 2) * B * H * 2 < (1ull << 32);
        const size_t lds_need = (size_t)(H / 16) * 1024 + (fused ? (size_t)(F / 16) * 1024 + 8192 : 0) + ASR_GLDS_BYTES + 4096;
        if (xsum_bf16) p.s_end = (T + 1) / 2;
        if (fits32 && pk[2] && launch_persist(pk, p, B, H, lds_need, ctl_words, err_flag, s)) {
            if (xsum_bf16) {        // the second half: same counters, state from hbuf / csave
                p.s_begin = p.s_end; p.s_end = T; p.xsum = (__bf16 *)xsum_bf16;
                if (!launch_persist(pks, p, B, H, lds_need, ctl_words, err_flag, s)) return ASR_ELAUNCH;
            }
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
        }
    }
    if (fused) return ASR_EUNSUPPORTED;
    // one launch per step.  64-row tiles (bt = 2) halve the W_hh re-reads but were
    // measured SLOWER (13.7 vs 11.4 us at B=512): the step is latency-bound per
    // workgroup, not fetch-bound
    const int bt = 1;
    const dim3 grid(H / 32, (B + 32 * bt - 1) / (32 * bt), 2);
    void (*kern)(LstmFwdParams) = nullptr;
#define ASR_PICK(KSV)                                                              \
    if (H == 16 * KSV) kern = gx_bf16 ? lstm_fwd_step_kernel<KSV, 1, 1> : lstm_fwd_step_kernel<KSV, 1, 0>;
    ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16) ASR_PICK(24) ASR_PICK(32) ASR_PICK(48)
#undef ASR_PICK
    if (!kern) return ASR_EUNSUPPORTED;       // hidden sizes built: 64,128,256,320,384,512,768
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
}  // namespace

namespace {
int lstm_bwd_impl(const float *dy, int dy_shared, const void *whhT_bf16, const void *wihT_bf16,
                  const int32_t *lens, int T, int B, int H, const void *gates_bf16,
                  const float *csave, void *dgates_bf16, float *dx, void *workspace,
                  int64_t workspace_bytes, uint32_t *err_flag, void *stream);
}

extern "C" int asr_lstm_bidir_bwd_bf16(const float *dy, int dy_shared, const void *whhT_bf16,
                                       const int32_t *lens, int T, int B, int H,
                                       const void *gates_bf16, const float *csave,
                                       void *dgates_bf16,
                                       void *workspace, int64_t workspace_bytes,
                                       uint32_t *err_flag, void *stream) {
    return lstm_bwd_impl(dy, dy_shared, whhT_bf16, nullptr, lens, T, B, H, gates_bf16, csave,
                         dgates_bf16, nullptr, workspace, workspace_bytes, err_flag, stream);
}

#ifdef ASR_EXPERIMENTS   // include/asr_amd_experiments.h: not in the default library
extern "C" int asr_lstm_bidir_bwd_fused_bf16(const float *dy, int dy_shared, const void *whhT_bf16,
                                             const void *wihT_bf16, const int32_t *lens,
                                             int T, int B, int H, const void *gates_bf16,
                                             const float *csave, void *dgates_bf16, float *dx,
                                             void *workspace, int64_t workspace_bytes,
                                             uint32_t *err_flag, void *stream) {
    if (!wihT_bf16 || !dx) return ASR_EINVAL;
    return lstm_bwd_impl(dy, dy_shared, whhT_bf16, wihT_bf16, lens, T, B, H, gates_bf16, csave,
                         dgates_bf16, dx, workspace, workspace_bytes, err_flag, stream);
}
#endif

namespace {
int lstm_bwd_impl(const float *dy, int dy_shared, const void *whhT_bf16, const void *wihT_bf16,
                  const int32_t *lens, int T, int B, int H, const void *gates_bf16,
                  const float *csave, void *dgates_bf16, float *dx, void *workspace,
                  int64_t workspace_bytes, uint32_t *err_flag, void *stream) {
    const bool fused = wihT_bf16 != nullptr;
    if (T < 0 || B <= 0 || H <= 0 || (H % 32) != 0 || dy_shared < 0 || dy_shared > 2) return ASR_EINVAL;
    if (T == 0) return ASR_OK;
    if (!dy || !whhT_bf16 || !lens || !gates_bf16 || !csave || !dgates_bf16 || !workspace)
        return ASR_EINVAL;
    if (fused && !(asr_lstm_fused_supported(B, H, H) & 2)) return ASR_EUNSUPPORTED;
    if (workspace_bytes < asr_lstm_workspace_bytes(B, H)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    LstmBwdParams p;
    const size_t Bp = (size_t)plane_rows(B);
    const size_t dbytes = (size_t)2 * 2 * Bp * 4 * H * 2, cbytes = (size_t)2 * B * H * 4;
    __bf16 *wpack = (__bf16 *)((char *)workspace + dbytes + cbytes);
    p.dy = dy; p.dy_shared = dy_shared; p.whhT = wpack; p.lens = lens;
    p.wihT = wpack + (size_t)2 * H * 4 * H; p.dx = dx;
    p.T = T; p.B = B; p.H = H;
    p.gates = (const u32x2 *)gates_bf16; p.csave = csave;
    p.dgbuf = (__bf16 *)workspace;
    p.dcbuf = (float *)((char *)workspace + dbytes);
    p.dgates = (__bf16 *)dgates_bf16;
    {
        ZeroList zl;
        zl.add(workspace, dbytes + cbytes);
        if (persist_enabled())
            zl.add((char *)workspace + asr_lstm_workspace_bytes(B, H) - ctl_bytes(B), (size_t)ctl_bytes(B));
        zl.launch(s);
    }
    // whhT_bf16 is [2][H][4H] row-major: rows = hidden unit j, cols = k over 4H
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, s,
                       (const __bf16 *)whhT_bf16, wpack, 2, H, 4 * H, 0);
    if (fused)          // wihT_bf16 is [2][H (input feature)][4H] row-major
        hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, s,
                           (const __bf16 *)wihT_bf16, wpack + (size_t)2 * H * 4 * H, 2, H, 4 * H, 0);
    if (persist_enabled()) {
        unsigned *ctl_words = (unsigned *)((char *)workspace + asr_lstm_workspace_bytes(B, H) - ctl_bytes(B));
        void (*pk[3])(LstmBwdParams, LstmTeamCtl) = {nullptr, nullptr, nullptr};
#define ASR_PICK(KSV) if (H == 16 * KSV && !fused) { pk[0] = lstm_bwd_persist_kernel<KSV, 2>; \
        pk[1] = lstm_bwd_persist_kernel<KSV, 3>; pk[2] = lstm_bwd_persist_kernel<KSV, 4>; }
        ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16) ASR_PICK(24)
#undef ASR_PICK
#define ASR_PICK(KSV) if (H == 16 * KSV && fused) { pk[0] = lstm_bwd_dx_kernel<KSV, 2>; \
        pk[1] = lstm_bwd_dx_kernel<KSV, 3>; }
        ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16)
#undef ASR_PICK
        // 32-bit byte offsets (the fused kernel marks masked lanes with offset 2^31: half the range)
        const bool fits32 = (uint64_t)T * B * 8 * H * 4 < (fused ? (1ull << 31) : (1ull << 32));
        const size_t lds = (size_t)(H / 4) * 1024 + ASR_GLDS_BYTES + 16384 + (fused ? 24 * 1280 : 0);
        if (fits32 && (pk[2] || fused) && launch_persist(pk, p, B, H, lds, ctl_words, err_flag, s))
            return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    // the per-step kernels know neither the fused input gradient nor the two-plane dy
    if (fused || dy_shared == 2) return ASR_EUNSUPPORTED;
    const int bt = 1;
    const dim3 grid(H / 32, (B + 32 * bt - 1) / (32 * bt), 2);
    void (*kern)(LstmBwdParams) = nullptr;
#define ASR_PICK(KSV)                                                              \
    if (H == 16 * KSV) kern = bt == 2 ? lstm_bwd_step_kernel<KSV, 2> : lstm_bwd_step_kernel<KSV, 1>;
    ASR_PICK(20) ASR_PICK(4) ASR_PICK(8) ASR_PICK(16) ASR_PICK(24) ASR_PICK(32) ASR_PICK(48)
#undef ASR_PICK
    if (!kern) return ASR_EUNSUPPORTED;
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
}  // namespace
