// Weight gradients of one bidirectional LSTM layer on gfx950 (include/asr_amd.h:
// asr_lstm_wgrad_bf16):
//
//     dW_ih[d] [4H, H] = dgates_dᵀ · x          dW_hh[d] [4H, H] = dgates_dᵀ · h_prev_d
//
// i.e. C[M, N] = Aᵀ·B with M = 8H gate rows (both directions), N = H and K = T·B frames —
// reductions over 192 k frames into 6.5 MB of output.  As library GEMMs (three batched
// products over chunks of frames + sums, native_lstm._weight_gradients_library) they run at
// 0.67 PFLOP/s: 0.94 ms per layer, 3.7 ms of a 20.6 ms training step at B = 576.  Here:
//
//  * a job is a [256 gate rows] x [320 columns of ONE B operand] block of C for one chunk of
//    frames: 10 row blocks x {x} + 5 x {h_prev_fwd} + 5 x {h_prev_rev} = 20 blocks x 12 chunks
//    = 240 workgroups, one per CU; wave (wm, wn) of a 4 x 2 grid accumulates 64 x 160 of it in
//    160 registers (2 x 5 tiles of `v_mfma_f32_32x32x16_bf16`): 10 MFMAs per k-step from 7
//    operand fragments;
//  * both operands are k-major in memory (a row = one frame), the layout
//    `ds_read_b64_tr_b16` transposes for free: tiles of 16 frames (one k-step, 18 KB) go
//    global -> LDS unchanged by LDS-DMA into a ring of eight stages (six in flight, counted
//    `vmcnt`, one `s_barrier` per k-step) and are read as MFMA fragments with the transposing
//    read one k-step ahead of the MFMAs that use them;
//  * LDS-DMA writes lane-linear, so rows cannot be padded against bank conflicts; the 16-byte
//    chunks of a row are XOR-swizzled on the SOURCE side instead (a lane fetches the chunk
//    that belongs at its LDS position): the four rows a transposing read touches land in four
//    different 16-bank windows (SQ_LDS_BANK_CONFLICT = 0; pitches 512 B / 640 B alone would
//    put them on one / two);
//  * within a k-step the fragment reads, address arithmetic and DMA issue of the NEXT steps
//    are interleaved between the MFMAs in fixed groups (sched_barrier): issued in front of
//    them they left the matrix pipe idle for 40 % of every step — all eight waves leave the
//    barrier together;
//  * the workgroups that read the same B tiles (one frame chunk, one direction) are placed on
//    one XCD; the partial blocks of the frame chunks are summed by a second, small kernel.
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
constexpr int MT = 256;                 // gate rows per workgroup
constexpr int KT = 16;                  // frames per stage = one MFMA k-step
constexpr int NS = 8;                   // ring slots
constexpr int PA = MT * 2, PB = HN * 2; // row pitches in LDS (bytes)
constexpr int ACH = MT / 8, BCH = HN / 8;      // 16-byte chunks per row
constexpr int A_BYTES = KT * PA, B_BYTES = KT * PB;
constexpr int STAGE = A_BYTES + B_BYTES;       // 18 KiB
constexpr int A_DMA = KT * ACH / 64, B_DMA = KT * BCH / 64;     // 8 + 10 LDS-DMAs of 1 KiB
constexpr int PW = 3;                   // DMAs per wave and stage (the 6 surplus ones move nothing)
constexpr int DUMMY = NS * STAGE;       // 1 KiB the surplus DMAs write zeros to
constexpr int LDS_BYTES = NS * STAGE + 1024;
constexpr int NMT = 8 * HN / MT;        // 10 row blocks

struct WgradParams {
    const __bf16 *dg;         // [K][8H]
    const __bf16 *x;          // [K][H] or null
    const __bf16 *hp[2];      // h_prev of direction 0 / 1: [K][H]
    float *partial;           // [kc][job][wave][tile][reg][lane]
    int K;                    // frames
    int rows_per_chunk;       // multiple of 2 * KT
    int njobs;                // 20 with x, 10 without
    int xcd_groups;           // (chunk, direction) groups per XCD, 0 = plain placement
};

// 64 lanes x 16 bytes global -> LDS at lds_byte + 16 * lane; out-of-range lanes write zeros
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
// MFMA fragment (8 consecutive k of the lane's row/column) from a k-major LDS image: two
// transposing reads of 4 k each, `pitch4` bytes (4 image rows) apart
__device__ __forceinline__ bf16x8 tr_frag(const char *addr, int pitch4) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr + pitch4));
    s16x8 v;
    v.s0 = lo.x; v.s1 = lo.y; v.s2 = lo.z; v.s3 = lo.w;
    v.s4 = hi.x; v.s5 = hi.y; v.s6 = hi.z; v.s7 = hi.w;
    return __builtin_bit_cast(bf16x8, v);
}

// Swizzles (chunk index within a row, by the row's k & 3): rows q = 0..3 of one transposing
// read must fall into four different 16-bank windows.
//   A, pitch 512 B (a multiple of the 256-B bank row): chunk ^ (q << 2)
//   B, pitch 640 B (= 2.5 bank rows: rows q and q+2 collide): chunk ^ ((q >> 1) << 2)
__device__ __forceinline__ int swz_a(int chunk, int k) { return chunk ^ ((k & 3) << 2); }
__device__ __forceinline__ int swz_b(int chunk, int k) { return chunk ^ (((k & 3) >> 1) << 2); }

__global__ __launch_bounds__(512) void lstm_wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // ---- job: (frame chunk kc, row block mt, B operand).  The 10 workgroups of one (chunk,
    // direction) — 5 row blocks x {x, h_prev of that direction} — read the same B tiles and
    // pairwise the same A tiles; they are put on ONE XCD (under the observed round-robin
    // dispatch workgroup w runs on XCD w % 8; xcd_groups groups per XCD).  Speed only.
    int kc, job;
    if (p.xcd_groups) {
        const int w = blockIdx.x, per = p.njobs / 2;             // workgroups per group
        const int slot = w >> 3, grp = (w & 7) * p.xcd_groups + slot / per;
        kc = grp >> 1;
        job = (grp & 1) * per + slot % per;
    } else {
        job = blockIdx.x % p.njobs;
        kc = blockIdx.x / p.njobs;
    }
    // with x: jobs [0,10) direction 0 (5 x-blocks, 5 h-blocks), [10,20) direction 1
    const int per_dir = p.njobs / 2, dir = job / per_dir, jj = job % per_dir;
    const bool is_x = p.x != nullptr && jj < 5;
    const int mt = dir * 5 + (p.x != nullptr ? jj % 5 : jj);
    const __bf16 *bsrc = is_x ? p.x : p.hp[dir];
    const int k_begin = kc * p.rows_per_chunk, nstage = p.rows_per_chunk / KT;

    // ---- loader: DMA i of this wave is number q = wave + 8 i of the stage's [A | B] list;
    // everything about it that does not change from stage to stage sits in SGPRs / one VGPR
    rsrc_words rs[PW];
    u32 voff[PW], lds_off[PW], soff[PW], stride[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave + 8 * i;
        const bool isa = q < A_DMA, real = q < A_DMA + B_DMA;
        const unsigned long long base = (unsigned long long)(isa ? (const void *)p.dg : (const void *)bsrc);
        rs[i].x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rs[i].y = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
        rs[i].z = __builtin_amdgcn_readfirstlane(real ? (int)((u32)p.K * (isa ? 8 * HN * 2 : HN * 2)) : 0);
        rs[i].w = 0x00020000;
        stride[i] = isa ? KT * 8 * HN * 2 : KT * HN * 2;
        soff[i] = (u32)k_begin * (isa ? 8 * HN * 2 : HN * 2);
        lds_off[i] = real ? (u32)q * 1024u : (u32)DUMMY;
        if (isa) {                        // LDS chunk L = 64 q + lane of the [KT][ACH] image
            const int L = 64 * q + lane, k = L / ACH, pos = L % ACH;
            voff[i] = (u32)k * (8 * HN * 2) + (u32)mt * PA + (u32)swz_a(pos, k) * 16u;
        } else {
            const int L = 64 * ((q - A_DMA) % B_DMA) + lane, k = L / BCH, pos = L % BCH;
            voff[i] = (u32)k * (HN * 2) + (u32)swz_b(pos, k) * 16u;
        }
    }
    u32 wslot = 0;                        // ring slot (byte offset) the next issued stage goes to
    const u32 smem_base = lds_addr(smem);
    auto issue_one = [&](int i) {
        dma16(rs[i], smem_base + wslot + lds_off[i], voff[i], soff[i]);
        soff[i] += stride[i];
    };
    auto issue_done = [&]() { wslot = wslot + STAGE == NS * STAGE ? 0u : wslot + STAGE; };

    // ---- consumer: lane parts of the transposing-read addresses
    // group g = lane >> 4: column sub-block g & 1, k-half g >> 1; lane 4 q + pp of the group
    // supplies row q, columns 4 pp .. 4 pp + 3 (cdna_hip_programming.md T10)
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (g >> 1) + q4;
    u32 a_off[2], b_off[5];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = (wm * 64 + 32 * i) / 8 + 2 * (g & 1) + (pp >> 1);
        a_off[i] = (u32)krow * PA + (u32)swz_a(chunk, krow) * 16u + (u32)(pp & 1) * 8u;
    }
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
    u32 rslot = 0;                        // ring slot of the stage whose fragments are read next

    // One k-step.  At its top stage st + 1 has landed (this wave's part: all but the DMAs of
    // the NS - 3 stages behind it; everyone's after the barrier, which also says every wave
    // has its fragments of stage st - 1 in registers, so that slot takes stage st + NS - 1).
    // Five groups of two MFMAs (stage st, operands fetched by the previous step); between
    // them the fragment reads of stage st + 1 and the three DMAs.
#define WG_MFMA(J) do { \
        acc[0][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[J], acc[0][J], 0, 0, 0); \
        acc[1][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[J], acc[1][J], 0, 0, 0); } while (0)
    auto step = [&](bf16x8 (&a)[2], bf16x8 (&b)[5], bf16x8 (&na)[2], bf16x8 (&nb)[5]) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 3) * PW) : "memory");
        const char *rs_ = smem + rslot;
        WG_MFMA(0);
        na[0] = tr_frag(rs_ + a_off[0], 4 * PA);
        na[1] = tr_frag(rs_ + a_off[1], 4 * PA);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(1);
        nb[0] = tr_frag(rs_ + b_off[0], 4 * PB);
        nb[1] = tr_frag(rs_ + b_off[1], 4 * PB);
        issue_one(0);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(2);
        nb[2] = tr_frag(rs_ + b_off[2], 4 * PB);
        nb[3] = tr_frag(rs_ + b_off[3], 4 * PB);
        issue_one(1);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(3);
        nb[4] = tr_frag(rs_ + b_off[4], 4 * PB);
        issue_one(2);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(4);
        issue_done();
        rslot = rslot + STAGE == NS * STAGE ? 0u : rslot + STAGE;
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: stages 0 .. NS-2 in flight, stage 0 landed, its fragments read
    for (int st = 0; st < NS - 1; ++st) {
#pragma unroll
        for (int i = 0; i < PW; ++i) issue_one(i);
        issue_done();
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * PW) : "memory");
    fa[0][0] = tr_frag(smem + a_off[0], 4 * PA);
    fa[0][1] = tr_frag(smem + a_off[1], 4 * PA);
#pragma unroll
    for (int j = 0; j < 5; ++j) fb[0][j] = tr_frag(smem + b_off[j], 4 * PB);
    rslot = STAGE;
    for (int st = 0; st < nstage; st += 2) {          // nstage is even
        step(fa[0], fb[0], fa[1], fb[1]);
        step(fa[1], fb[1], fa[0], fb[0]);
    }
#undef WG_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stages requested past the chunk
    // ---- partial block, raw accumulator order [tile][reg][lane]
    float *out = p.partial + ((((size_t)kc * p.njobs + job) * 8 + wave) * 10) * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((i * 5 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
}

// sum of the nkc partial blocks, written to dW_ih [8H][H] / dW_hh [2][4H][H]
__global__ void lstm_wgrad_reduce_kernel(const float *partial, int nkc, int njobs, int with_x,
                                         float *dw_ih, float *dw_hh) {
    const size_t per = (size_t)njobs * 8 * 10 * 1024;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    float s = 0.f;
    for (int kc = 0; kc < nkc; ++kc) s += partial[(size_t)kc * per + idx];
    const int lane = (int)(idx & 63), reg = (int)((idx >> 6) & 15);
    size_t rest = idx >> 10;
    const int tile = (int)(rest % 10); rest /= 10;
    const int wave = (int)(rest & 7), job = (int)(rest >> 3);
    const int per_dir = njobs / 2, dir = job / per_dir, jj = job % per_dir;
    const bool is_x = with_x && jj < 5;
    const int mt = dir * 5 + (with_x ? jj % 5 : jj);
    const int wm = wave >> 1, wn = wave & 1, i = tile / 5, j = tile % 5;
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int m = mt * MT + wm * 64 + 32 * i + row;
    const int n = wn * 160 + 32 * j + (lane & 31);
    (is_x ? dw_ih : dw_hh)[(size_t)m * HN + n] = s;
}

inline int cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

struct Plan { int njobs, nkc, rows_per_chunk, xcd_groups; };
inline Plan make_plan(int K, int with_x) {
    Plan pl;
    pl.njobs = with_x ? 2 * NMT : NMT;
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    pl.nkc = cus / pl.njobs > 0 ? cus / pl.njobs : 1;
    // whole (chunk, direction) groups per XCD: 8 XCDs x groups x njobs/2 workgroups
    const int per_xcd = cus / 8, half = pl.njobs / 2;
    pl.xcd_groups = (cus % 8 == 0 && per_xcd >= half) ? per_xcd / half : 0;
    if (pl.xcd_groups) pl.nkc = 8 * pl.xcd_groups / 2;
    if (pl.nkc < 1) { pl.nkc = 1; pl.xcd_groups = 0; }
    const int per = (K + pl.nkc - 1) / pl.nkc;
    pl.rows_per_chunk = (per + 2 * KT - 1) / (2 * KT) * (2 * KT);        // an even number of stages
    if (pl.rows_per_chunk < 2 * KT) pl.rows_per_chunk = 2 * KT;
    return pl;
}

}  // namespace

extern "C" int asr_lstm_wgrad_supported(int H) { return H == HN ? 1 : 0; }

extern "C" int64_t asr_lstm_wgrad_workspace_bytes(int T, int B, int H, int with_input) {
    if (T < 0 || B < 0 || H != HN) return -1;
    const Plan pl = make_plan(T * B, with_input);
    return (int64_t)pl.nkc * pl.njobs * 8 * 10 * 1024 * 4;
}

extern "C" int asr_lstm_wgrad_bf16(const void *dgates_bf16, const void *x_bf16, const void *y_bf16,
                                   int T, int B, int H, float *dw_ih, float *dw_hh,
                                   void *workspace, int64_t workspace_bytes, void *stream) {
    if (T <= 0 || B <= 0 || !dgates_bf16 || !y_bf16 || !dw_hh || !workspace) return ASR_EINVAL;
    if (H != HN) return ASR_EUNSUPPORTED;
    if ((x_bf16 == nullptr) != (dw_ih == nullptr)) return ASR_EINVAL;
    const int with_x = x_bf16 ? 1 : 0;
    const int64_t K = (int64_t)T * B;
    // 32-bit buffer offsets, with room for the stages requested past the last chunk
    if ((K + 64 * KT) * 8 * HN * 2 >= (1ll << 31)) return ASR_EUNSUPPORTED;
    if (workspace_bytes < asr_lstm_wgrad_workspace_bytes(T, B, H, with_x)) return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const Plan pl = make_plan((int)K, with_x);
    WgradParams p;
    p.dg = (const __bf16 *)dgates_bf16;
    p.x = (const __bf16 *)x_bf16;
    // h_{t-1}: the forward direction looks one frame back (frames 0..T-1 of its zero-padded
    // plane [T+2,B,H]), the reverse one frame ahead (frames 2..T+1 of its plane)
    const __bf16 *y = (const __bf16 *)y_bf16;
    p.hp[0] = y;
    p.hp[1] = y + ((size_t)(T + 2) + 2) * B * HN;
    p.partial = (float *)workspace;
    p.K = (int)K;
    p.rows_per_chunk = pl.rows_per_chunk;
    p.njobs = pl.njobs;
    p.xcd_groups = pl.xcd_groups;
    const size_t per = (size_t)pl.njobs * 8 * 10 * 1024;
    if (hipFuncSetAttribute((const void *)lstm_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
        return ASR_ELAUNCH;
    hipLaunchKernelGGL(lstm_wgrad_kernel, dim3(pl.njobs * pl.nkc), dim3(512), LDS_BYTES, s, p);
    hipLaunchKernelGGL(lstm_wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                       p.partial, pl.nkc, pl.njobs, with_x, dw_ih, dw_hh);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
