// Input gradient of one bidirectional LSTM layer on gfx950 (include/asr_amd.h:
// asr_lstm_dgrad_bf16):   dx [T*B, H] = dgates [T*B, 8H] · W_ih [8H, H]
// (both directions in one product: K = 8H).  As a library GEMM (fp32 output: outside
// TunableOp) 0.37 ms per layer at B = 576, 0.85 PFLOP/s.  Here, with the machinery of
// lstm_wgrad.hip:
//  * a workgroup (8 waves as 4 x 2) owns 256 frames x all 320 columns: 160 accumulator
//    registers per lane, 10 MFMAs per k-step from 7 fragments; K = 2560 in 40 stages of 64;
//  * A = dgates is row-major (k contiguous): a stage [256 rows x 64 k] goes global -> LDS by
//    LDS-DMA in whole 128-byte row pieces, chunk-swizzled on the source side
//    (chunk ^ ((row >> 1) & 7): the 16 rows of a `ds_read_b128` lane group then cover one
//    256-byte bank row exactly), fragments by `ds_read_b128`;
//  * B = W_ih is k-major (a row = one gate unit): stages [64 k x 320] by LDS-DMA, fragments by
//    the transposing `ds_read_b64_tr_b16`, swizzled as in lstm_wgrad.hip; the 1.6 MB of W_ih
//    are re-streamed from L2 by every workgroup;
//  * two 72 KB ring slots; within a stage the fragments of k-step s+1 are read under the
//    MFMAs of k-step s; the barrier that hands over the next stage sits in front of the last
//    k-step of the current one, so the first fragments of the next stage are read under it.
//    (Four 36 KB slots with two stages in flight and 64-byte row pieces: 356 us against 335.)
//  Ablations at B = 576 (339 us): without the DMAs 240, without the MFMAs 290 — the memory
//  side (2.2 GB of LDS-DMA per call, 55 % of it W_ih from L2) bounds it at ~9 TB/s aggregate.
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int rsrc_words;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int u32;

constexpr int HN = 320;                 // hidden size this kernel is built for
constexpr int KD = 8 * HN;              // reduction length (both directions' gates)
constexpr int MT = 256;                 // frames per workgroup
constexpr int KS = 64;                  // k per stage (4 MFMA k-steps)
constexpr int A_BYTES = MT * KS * 2;    // 32 KiB: [256 rows][8 chunks of 16 B]
constexpr int PB = HN * 2, BCH = HN / 8;
constexpr int B_BYTES = KS * PB;        // 40 KiB: [64 k][40 chunks]
constexpr int STAGE = A_BYTES + B_BYTES;
constexpr int A_DMA = A_BYTES / 1024, B_DMA = B_BYTES / 1024;      // 32 + 40
constexpr int A_PW = A_DMA / 8, B_PW = B_DMA / 8;                   // 4 + 5 per wave
constexpr int NSTAGE = KD / KS;         // 40
constexpr int LDS_BYTES = 2 * STAGE;

struct DgradParams {
    const __bf16 *dg;         // [M][KD]
    const __bf16 *w;          // [KD][HN]
    float *dx;                // [M][HN]
    int M;
};

__device__ __forceinline__ rsrc_words raw_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    rsrc_words r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *ptr) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const void *)ptr);
}
__device__ __forceinline__ bf16x8 tr_frag(const char *addr, int pitch4) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr + pitch4));
    s16x8 v;
    v.s0 = lo.x; v.s1 = lo.y; v.s2 = lo.z; v.s3 = lo.w;
    v.s4 = hi.x; v.s5 = hi.y; v.s6 = hi.z; v.s7 = hi.w;
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int swz_a(int chunk, int row) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_b(int chunk, int k) { return chunk ^ (((k & 3) >> 1) << 2); }

__global__ __launch_bounds__(512) void lstm_dgrad_kernel(DgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * MT;

    // ---- loader
    const rsrc_words rA = raw_rsrc(p.dg, (u32)p.M * (KD * 2));
    const rsrc_words rB = raw_rsrc(p.w, (u32)KD * PB);
    u32 va[A_PW], vb[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {      // LDS chunk L of the [256][8] image holds chunk pos ^ f(row)
        const int L = 64 * (wave + 8 * i) + lane, row = L >> 3, pos = L & 7;
        va[i] = (u32)(m0 + row) * (KD * 2) + (u32)swz_a(pos, row) * 16u;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        const int L = 64 * (wave + 8 * i) + lane, k = L / BCH, pos = L % BCH;
        vb[i] = (u32)k * PB + (u32)swz_b(pos, k) * 16u;
    }
    const u32 smem_base = lds_addr(smem);
    auto issue = [&](int s) {             // stage s -> slot s & 1; past the end: far out of range (zeros)
        const u32 slot = smem_base + (u32)(s & 1) * STAGE;
        const u32 soA = s < NSTAGE ? (u32)s * (KS * 2) : 0x7ff00000u;
        const u32 soB = s < NSTAGE ? (u32)s * (KS * PB) : 0x7ff00000u;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) dma16(rA, slot + (u32)(wave + 8 * i) * 1024u, va[i], soA);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) dma16(rB, slot + A_BYTES + (u32)(wave + 8 * i) * 1024u, vb[i], soB);
    };

    // ---- consumer addresses (lane parts)
    // A fragment of rows wm*64 + 32 i + (lane & 31), k-step ks: chunk 2 ks + (lane >> 5)
    const int arow = wm * 64 + (lane & 31);
    u32 a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        a_off[ks] = (u32)arow * 128u + (u32)swz_a(2 * ks + (lane >> 5), arow) * 16u;     // (+ 32 rows: + 4096,
                                                                                         //  same swizzle: 32 >> 1 & 7 = 0)
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (g >> 1) + q4;
    u32 b_off[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int chunk = (wn * 160 + 32 * j) / 8 + 2 * (g & 1) + (pp >> 1);
        b_off[j] = (u32)A_BYTES + (u32)krow * PB + (u32)swz_b(chunk, krow) * 16u + (u32)(pp & 1) * 8u;
    }
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][2], fb[2][5];
    auto read_frags = [&](int slot, int ks, bf16x8 (&a)[2], bf16x8 (&b)[5]) {
        const char *sl = smem + (size_t)slot * STAGE;
        a[0] = *reinterpret_cast<const bf16x8 *>(sl + a_off[ks]);
        a[1] = *reinterpret_cast<const bf16x8 *>(sl + a_off[ks] + 4096);
#pragma unroll
        for (int j = 0; j < 5; ++j) b[j] = tr_frag(sl + b_off[j] + ks * 16 * PB, 4 * PB);
    };
    auto mfma10 = [&](bf16x8 (&a)[2], bf16x8 (&b)[5]) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(A_PW + B_PW) : "memory");
    read_frags(0, 0, fa[0], fb[0]);
    for (int s = 0; s < NSTAGE; ++s) {
        const int sl = s & 1;
        read_frags(sl, 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma10(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(sl, 2, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfma10(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(sl, 3, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma10(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        // every fragment of this stage is in registers (lgkmcnt(0)) and stage s + 1 has landed
        // (this wave's DMAs: vmcnt(0)); after the barrier that holds for all waves: the slot
        // takes stage s + 2 and the first fragments of stage s + 1 are read under the last MFMAs
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(s + 2);
        read_frags(sl ^ 1, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfma10(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- dx rows: C layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    p.dx[(size_t)row * HN + wn * 160 + 32 * j + (lane & 31)] = acc[i][j][r];
            }
        }
}

}  // namespace

extern "C" int asr_lstm_dgrad_supported(int H) { return H == HN ? 1 : 0; }

extern "C" int asr_lstm_dgrad_bf16(const void *dgates_bf16, const void *w_ih_bf16, int T, int B, int H,
                                   float *dx, void *stream) {
    if (T <= 0 || B <= 0 || !dgates_bf16 || !w_ih_bf16 || !dx) return ASR_EINVAL;
    if (H != HN) return ASR_EUNSUPPORTED;
    const int64_t M = (int64_t)T * B;
    if (M * KD * 2 >= (1ll << 31) - (1 << 20)) return ASR_EUNSUPPORTED;      // 32-bit buffer offsets
    DgradParams p;
    p.dg = (const __bf16 *)dgates_bf16;
    p.w = (const __bf16 *)w_ih_bf16;
    p.dx = dx;
    p.M = (int)M;
    if (hipFuncSetAttribute((const void *)lstm_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
        return ASR_ELAUNCH;
    hipLaunchKernelGGL(lstm_dgrad_kernel, dim3((unsigned)((M + MT - 1) / MT)), dim3(512), LDS_BYTES,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
