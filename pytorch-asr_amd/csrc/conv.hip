// 7x7, 32 -> 32 channel convolution of the DeepSpeech2 front-end (reference
// att_speech/modules/encoders/deep_speech_2.py:60-73: Conv2d(32, 32, (7, 7), stride (3, 1)),
// 97 % of the conv stack's flops) as hand-written MFMA kernels on channels-last bf16:
// forward, input gradient, weight gradient.  fp32 accumulation, bias-free (the bias is folded
// into the fused BatchNorm kernels, csrc/bnact.hip).
//
// Layouts: x [B, H, W, 32] bf16 (H = time, W = frequency), y [B, Ho, Wo, 32] bf16 with
// Ho = (H - 7) / SH + 1, Wo = W - 6; w [32 co][32 ci][7][7] fp32 (nn.Conv2d's parameter).
//
// Common structure: a workgroup stages the input rows one block of output rows needs in LDS
// (80-byte pixel stride: 64 B of channels + 16 B pad, so the 16-byte MFMA fragment reads of 16
// consecutive pixels hit 16 disjoint bank groups), keeps its share of the weight fragments in
// REGISTERS for its whole life, and feeds v_mfma_f32_32x32x16_bf16 with one ds_read_b128 per
// MFMA whose address is a per-lane pixel base + an immediate (tap) offset.
#include <type_traits>
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int CH = 32;            // channels in and out
constexpr int KS = 7;             // kernel size (both axes)
constexpr int PIX = 80;           // bytes per pixel in LDS
constexpr int KSTEPS = KS * KS * CH / 16;      // 98 MFMA k-steps of 16

// LDS-DMA: 64 lanes x 16 bytes from a bounds-checked raw buffer straight to LDS at
// `lds_byte` + 16 * lane (out-of-range lanes write zeros); inline asm because hipcc orders
// every LDS read behind an LDS-DMA it knows of with s_waitcnt vmcnt(0)
typedef __attribute__((ext_vector_type(4))) int rsrc_words;
__device__ __forceinline__ rsrc_words conv_raw_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    rsrc_words r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void conv_dma16(rsrc_words r, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ unsigned conv_lds_addr(const void *ptr) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const void *)ptr);
}

// LDS row pitch (bytes) of a staged image whose MFMA tiles take 32 consecutive pixels of rows
// `per_row` pixels long, `step` image rows apart: the pitch >= min_pitch (multiple of 16, at
// most 256 bytes more) with the fewest ds_read_b128 bank conflicts over the first `ntiles`
// tiles.  ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) of a
// wave in one LDS cycle each when their 16 x 16 bytes fall into 64 distinct banks
// (MI355X_MICROARCH.md, LDS): a plain pitch of per_row * 80 bytes costs 2x on these shapes
// (rocprofv3: SQ_LDS_BANK_CONFLICT = 41 % of SQ_LDS_IDX_ACTIVE), the picked one 1x.
inline int pick_row_pitch(int min_pitch, int per_row, int step, int npix, int ntiles) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int best = min_pitch, best_cost = 1 << 30;
    for (int P = min_pitch; P <= min_pitch + 256; P += 16) {
        int cost = 0;
        for (int tile = 0; tile < ntiles; ++tile)
            for (int g = 0; g < 2; ++g) {
                int addr_of_bank[64], worst = 1, cnt[64];
                for (int b = 0; b < 64; ++b) { addr_of_bank[b] = -1; cnt[b] = 0; }
                for (int i = 0; i < 16; ++i) {
                    int m = 32 * tile + grp[g][i];
                    if (m >= npix) m = npix - 1;
                    const int r = m / per_row, w = m - r * per_row;
                    const int a = r * step * P + w * PIX;
                    for (int k = 0; k < 4; ++k) {
                        const int b = (a / 4 + k) & 63;
                        if (addr_of_bank[b] != a + 4 * k) {         // a new address on this bank
                            addr_of_bank[b] = a + 4 * k;
                            if (++cnt[b] > worst) worst = cnt[b];
                        }
                    }
                }
                cost += worst;
            }
        if (cost < best_cost) { best_cost = cost; best = P; }
    }
    return best;
}

inline int conv_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

// ---- per-channel statistics of an output image in LDS -------------------------------------
// oimg [npix][32] bf16 -> partial[64] = (sum over pixels, sum of squares) per channel, by
// 256 threads (8 pixel groups x 32 channels); `red` = 512 floats of LDS scratch
__device__ __forceinline__ void chan_partial_sums(const __bf16 *oimg, int npix, float *red, float *partial) {
    const int tid = threadIdx.x, ch = tid & 31, grp = tid >> 5;
    float s = 0.f, q = 0.f;
    // (eight reads in flight: one by one the loop pays an LDS round trip per pixel — 110 us of
    // the 357 us forward kernel at 38-wide rows)
    int m = grp;
    for (; m + 56 < npix; m += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (float)oimg[(m + 8 * u) * CH + ch];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += v[u];
            q += v[u] * v[u];
        }
    }
    for (; m < npix; m += 8) {
        const float v = (float)oimg[m * CH + ch];
        s += v;
        q += v * v;
    }
    red[grp * 64 + ch] = s;
    red[grp * 64 + 32 + ch] = q;
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 64 + tid];
        partial[tid] = t;
    }
}

// chan_sums[e] = sum over the workgroups' partial[g][e], e < 64, in double
__global__ __launch_bounds__(1024) void chan_sums_reduce_kernel(const float *partial, int nparts, double *out) {
    __shared__ double lds[1024];
    const int el = threadIdx.x & 63, way = threadIdx.x >> 6;
    double s = 0.0;
    for (int g = blockIdx.x * 16 + way; g < nparts; g += gridDim.x * 16) s += (double)partial[(size_t)g * 64 + el];
    lds[way * 64 + el] = s;
    __syncthreads();
    if (way == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += lds[w * 64 + el];
        atomicAdd(out + el, t);
    }
}

__global__ void zero_chan_sums_kernel(double *out) { out[threadIdx.x] = 0.0; }

// ---- weight packing -------------------------------------------------------------------
// forward: B[k][n], k = (kt * 7 + kf) * 32 + ci, n = co.
// fragment of k-step s for lane l: B[16 s + 8 (l >> 5) + j][l & 31], j = 0..7
__global__ void conv_pack_fwd_kernel(const float *w, __bf16 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KSTEPS * 64 * 8) return;
    const int j = i & 7, l = (i >> 3) & 63, s = i >> 9;
    const int k = 16 * s + 8 * (l >> 5) + j, co = l & 31;
    const int tap = k / CH, ci = k % CH, kt = tap / KS, kf = tap % KS;
    out[i] = (__bf16)w[((co * CH + ci) * KS + kt) * KS + kf];
}

// Diagnostic build only (-DCONV_STAMPS, ASR_CONV_STAMPS=1): per-phase s_memtime stamps of one
// workgroup's item loop, printed by the host entry points
#ifdef CONV_STAMPS
#define STAMP(k) do { if (p.stamps && blockIdx.x == 7 && lane == 0 && nst < 64) { p.stamps[(wave * 64 + nst) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct ConvFwdParams {
    long long *stamps;
    const __bf16 *x;
    const __bf16 *wpack;
    __bf16 *y;
    float *stats;                 // [workgroups][4 waves][64] channel sums of the outputs, or null
    int B, H, W, Ho, Wo, R;       // R output rows per workgroup (R * Wo <= FWD_PIX)
    int pitch;                    // bytes per image row in LDS (pick_row_pitch)
};

// M-tiles per wave of the forward kernel.  Three (192 pixels per workgroup) left no register
// for a second A fragment next to the 49 weight fragments and 48 accumulators: every MFMA
// waited for the LDS read issued right in front of it (ds_read -> s_waitcnt lgkmcnt(0) ->
// v_mfma, 147 times: MFMA-busy 19 %).  With two, three fragments are in flight.
constexpr int FWD_NT = 2, FWD_PIX = 2 * FWD_NT * 32, FWD_AD = 3;
constexpr int FWD_MAXQ = 15;      // DMA slots per wave (the image has at most 4 * FWD_MAXQ KiB)

// 64 lanes x 16 bytes to a bounds-checked raw buffer (out-of-range lanes are dropped): always
// exactly one VMEM instruction, so the `s_waitcnt vmcnt(N)` counted around it stays exact.
// (The builtin, not inline asm: a first asm version had its data registers overwritten by the
// next VALU instruction before the store had read them — 2 wait states the compiler inserts
// only for stores it can see.)
__device__ __forceinline__ void conv_store16(const void *base, unsigned bytes, u32x4 v, unsigned voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}

// one k-half (49 k-steps) of the product for FWD_NT M-tiles: the A fragments are read FWD_AD
// MFMAs ahead into a rotating register set, the order pinned (sched_group_barrier: the
// scheduler otherwise sinks every read next to its use again).  Operands swapped (weights as
// the MFMA's A, pixels as its B): a lane ends up with 4 consecutive co of one pixel per
// register quad.  The first k-step starts the accumulators from a literal zero.
template <int SH, int KH>
__device__ __forceinline__ void conv_fwd_half(const char *img, const unsigned (&pixbase)[FWD_NT], int pitch,
                                              const bf16x8 (&bf)[49], f32x16 (&acc)[FWD_NT]) {
    constexpr int N = 49 * FWD_NT;
    auto frag = [&](int idx) {
        const int s = idx / FWD_NT, i = idx % FWD_NT;
        const int ks = 49 * KH + s;
        const int kt = ks / 14, rem = ks % 14, kf = rem >> 1, cp = rem & 1;
        const unsigned off = (unsigned)(kt * pitch + kf * PIX + cp * 32);
        return *reinterpret_cast<const bf16x8 *>(img + pixbase[i] + off);
    };
    bf16x8 a[FWD_AD];
#pragma unroll
    for (int d = 0; d < FWD_AD; ++d) a[d] = frag(d);
#pragma unroll
    for (int idx = 0; idx < N; ++idx) {
        acc[idx % FWD_NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[idx / FWD_NT], a[idx % FWD_AD],
                                                                    idx < FWD_NT ? f32x16{} : acc[idx % FWD_NT], 0, 0, 0);
        if (idx + FWD_AD < N) a[idx % FWD_AD] = frag(idx + FWD_AD);
        // (k-step by k-step: the group barriers below fix how many reads and MFMAs alternate, not
        //  WHICH — left alone the scheduler finishes one tile first and parks the other's
        //  fragments, then the weights, in scratch)
        // (not behind the last k-step: the asm would stand between the MFMAs and the first reader
        //  of their results, and the wait states that reader needs are counted from its producer)
        if (idx % FWD_NT == FWD_NT - 1 && idx + 1 < N) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
    }
#pragma unroll
    for (int d = 0; d < FWD_AD; ++d) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int idx = 0; idx < N; ++idx) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (idx + FWD_AD < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

// One instantiation per k-half (the wave's role; see conv_dgrad_wave).  Per item a wave spends
// 98 MFMAs (3.1 k cycles); the first persistent version spent 10 k more per item on staging
// with run-time divisions (5.2 k), a serialized exchange of the k-halves (2.6 k) and the
// copy-out (1 k).  Now:
//   * the item-invariant part of a wave's <= 15 DMA slot offsets is worked out once and parked
//     in LDS (the 196 weight registers leave no room): per item a table read, one add and
//     the DMA; the buffer descriptor spans one utterance, rows past it read out of range;
//   * the next item's image is in flight under the epilogue (its own LDS area);
//   * the two waves of a pixel half finish one tile each: the other tile's partial sums go
//     through LDS (4 x 16 bytes per lane each way), every wave converts, transposes (LDS,
//     8-byte writes / 16-byte reads, chunks XOR-swizzled), sums the statistics of and stores
//     its OWN 32 pixels: two barriers per item.
template <int SH, int KH>
__device__ __forceinline__ void conv_fwd_wave(const ConvFwdParams &p, char *smem, int lane, int wave) {
    const int mh = wave >> 1;
    const int W = p.W, Wo = p.Wo, R = p.R;
    const int in_rows = SH * (R - 1) + KS;
    const unsigned img_bytes = (unsigned)in_rows * (unsigned)p.pitch;
    const int nslots = (int)((img_bytes + 1023u) / 1024u);
    // LDS: [image: nslots KiB | slot table: nslots x 64 u16 | exchange: 4 waves x 4 KiB]
    unsigned short *tab = reinterpret_cast<unsigned short *>(smem + nslots * 1024);
    char *xch = smem + nslots * (1024 + 128);
    // DMA slot q (this wave's: q = wave, wave + 4, ...) = LDS bytes [1024 q + 16 lane, +16) of the
    // linear image (rows `pitch` apart, pixels 80 bytes apart): 16-byte index of its x bytes from
    // the item's first input row, or 0xffff for padding.  Written and read by the same lane.
#pragma unroll 1
    for (int q = wave; q < nslots; q += 4) {
        const unsigned beta = (unsigned)q * 1024u + (unsigned)lane * 16u;
        const unsigned row = beta / (unsigned)p.pitch, rem = beta - row * (unsigned)p.pitch;
        const unsigned pix = (rem * 52429u) >> 22, within = rem - pix * 80u;       // rem / 80 for rem < 2^16
        const bool ok = beta < img_bytes && pix < (unsigned)W && within < 64u;
        tab[q * 64 + lane] = (unsigned short)(ok ? (row * (unsigned)W + pix) * 4u + (within >> 4) : 0xffffu);
    }
    // ---- per-lane pixel bases of this wave's M-tiles -------------------------------------------
    unsigned pixbase[FWD_NT];
    const int npix = R * Wo;
#pragma unroll
    for (int i = 0; i < FWD_NT; ++i) {
        int m = 32 * (FWD_NT * mh + i) + (lane & 31);
        if (m >= npix) m = npix - 1;                            // computed, never stored
        const int r = m / Wo, wo = m - r * Wo;
        pixbase[i] = (unsigned)((r * SH) * p.pitch + wo * PIX + (lane >> 5) * 16);
    }
    const int tiles = (p.Ho + R - 1) / R, nitems = tiles * p.B;
    const unsigned img_lds = conv_lds_addr(smem);
    const unsigned utt_bytes = (unsigned)(p.H * W * 64);
    auto stage = [&](int it, int ln) {
        const int b = it / tiles, ho0 = (it - b * tiles) * R;
        const rsrc_words xR = conv_raw_rsrc(reinterpret_cast<const char *>(p.x) + (size_t)b * utt_bytes, utt_bytes);
        const unsigned base = (unsigned)(ho0 * SH * W * 64);
        // (five slots per LDS round trip, fenced: the compiler otherwise gathers every independent
        //  LDS read of the epilogue at its top — more registers than the weights leave)
#pragma unroll
        for (int k0 = 0; k0 < FWD_MAXQ; k0 += 5) {
            unsigned t[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (wave + 4 * (k0 + k) < nslots) t[k] = tab[(wave + 4 * (k0 + k)) * 64 + ln];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (wave + 4 * (k0 + k) < nslots)
                    conv_dma16(xR, (unsigned)__builtin_amdgcn_readfirstlane((int)(img_lds + (unsigned)(wave + 4 * (k0 + k)) * 1024u)),
                               t[k] == 0xffffu ? 0x80000000u : t[k] * 16u + base, 0u);
            asm volatile("" ::: "memory");
        }
    };
    if ((int)blockIdx.x < nitems) stage(blockIdx.x, lane);
    // epilogue constants (see conv_dgrad_wave)
    const int partner = wave ^ 1, tile = FWD_NT * mh + KH;      // the M-tile this wave finishes
    char *give = xch + wave * 4096, *take = xch + partner * 4096;
    // running channel statistics of this wave's tiles, per lane (channel lane & 31, pixels
    // 16 (lane >> 5) ..): parked in LDS like everything else that would be live across the MFMA loop
    float *run = reinterpret_cast<float *>(xch + 4 * 4096) + wave * 128;
    run[lane] = 0.f;
    run[64 + lane] = 0.f;
    // ---- weight fragments of this wave's k-half: loaded ONCE per workgroup.  The grid is
    // persistent (two workgroups per CU, each walking the (utterance, row block) items): with
    // one item per workgroup every workgroup re-read its 196 KB of fragments from L2 — 3.5 GB
    // per call against 0.6 GB of input, and at the ~25 B/clk a CU takes in that, not the MFMA
    // loop, set the pace (469 us with or without the pipelined loop above).
    bf16x8 bf[49];
    {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p.wpack) + (size_t)(49 * KH) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 49; ++s) bf[s] = src[(size_t)s * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int nst = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++nst) {
        const int b = item / tiles, ho0 = (item - b * tiles) * R;
        STAMP(0);
        // the image's DMAs are older than the previous item's two stores (VMEM retires in order)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();                // image landed; the previous item's epilogue is done with LDS
        STAMP(1);
        f32x16 acc[FWD_NT];
        // (scheduling fences: without them the scheduler finishes the tile that is given away
        //  first, reads a dozen A fragments ahead to do so and spills the weights to make room)
        __builtin_amdgcn_sched_barrier(0);
        conv_fwd_half<SH, KH>(smem, pixbase, p.pitch, bf, acc);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        // (the lane index is worked out again here, per item: the weights, the accumulators and
        //  the fragments in flight leave the MFMA loop no register for anything else)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x16 &c = acc[1 - KH];
            *reinterpret_cast<f32x4 *>(give + (g * 64 + ln) * 16) = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
        STAMP(3);
        __syncthreads();                // image no longer read; the partial sums are in LDS
        STAMP(4);
        {
            const unsigned wr_off = (unsigned)((ln & 31) * 64 + ((ln >> 5) << 3));
            const unsigned wr_swz = (unsigned)((ln >> 2) & 3);
            f32x4 o[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) o[g] = *reinterpret_cast<const f32x4 *>(take + (g * 64 + ln) * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (reads before the writes below: same LDS)
            STAMP(5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16 &c = acc[KH];
                f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                v = KH == 0 ? v + o[g] : o[g] + v;              // k-half 0 + k-half 1 on both sides
                bf16x4 o4 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4 *>(take + wr_off + (((unsigned)g ^ wr_swz) << 4)) = o4;
            }
        }
        // (behind the exchange, not in front of it: with the accumulators still live the slot
        //  offsets do not fit next to the 196 weight registers)
        asm volatile("" ::: "memory");
        STAMP(6);
        if (item + (int)gridDim.x < nitems) stage(item + gridDim.x, ln);
        const int rows_here = (p.Ho - ho0) < R ? (p.Ho - ho0) : R;
        const int nvalid = rows_here * Wo - 32 * tile;           // this tile's pixels that exist
        if (p.stats) {
            const unsigned ch = (unsigned)ln & 31u;
            const unsigned p0 = (unsigned)(ln >> 5) * 16u;
            float ssum = 0.f, qsum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned pp = p0 + (unsigned)k;
                const unsigned short raw = *reinterpret_cast<const unsigned short *>(
                    take + pp * 64u + ((((ch >> 3) ^ (pp >> 2)) & 3u) << 4) + (ch & 7u) * 2u);
                const float v = (int)pp < nvalid ? __uint_as_float((unsigned)raw << 16) : 0.f;
                ssum += v;
                qsum += v * v;
                if ((k & 3) == 3) asm volatile("" ::: "memory");
            }
            run[ln] += ssum;
            run[64 + ln] += qsum;
        }
        asm volatile("" ::: "memory");
        STAMP(7);
        {
            // the tile's 2 KiB = bytes [2048 tile, +2048) of the item's output rows
            const char *ybase = reinterpret_cast<const char *>(p.y) + ((size_t)b * p.Ho + ho0) * Wo * 64;
            u32x4 v[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const unsigned e = (unsigned)(ln + 64 * it);
                v[it] = *reinterpret_cast<const u32x4 *>(take + ((e ^ ((e >> 4) & 3u)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 2; ++it)
                conv_store16(ybase, (unsigned)(rows_here * Wo * 64), v[it], (unsigned)(2048 * tile + 16 * (ln + 64 * it)));
        }
    }
    if (p.stats) {
        // per-wave partial sums: lanes 0..31 the channel sums, 32..63 the sums of squares
        const int ln = threadIdx.x & 63;
        const float a0 = run[ln & 31] + run[32 + (ln & 31)], a1 = run[64 + (ln & 31)] + run[96 + (ln & 31)];
        p.stats[((size_t)blockIdx.x * 4 + wave) * 64 + ln] = ln < 32 ? a0 : a1;
    }
}

template <int SH>
__global__ __launch_bounds__(256, 2) void conv7x7c32_fwd_kernel(ConvFwdParams p) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((wave & 1) == 0) conv_fwd_wave<SH, 0>(p, smem, lane, wave);
    else conv_fwd_wave<SH, 1>(p, smem, lane, wave);
}


// =========================================================================================
// input gradient, stride (3, 1):  dx[b, 3q + r, w, ci] = sum_{j, kf, co} dy[b, q - j, w - kf, co]
//                                                        * w[co, ci, r + 3j, kf]
// i.e. per row class r = h mod 3 a stride-1 correlation of dy (zero-padded by 6 pixels left and
// right and 2 rows above) with a J_r x 7 kernel, J_0 = 3, J_1 = J_2 = 2.
// B[k][n]: k = (j, kf, co) -> 16-steps (j, kf, co half), n = ci.  Packed per class:
// class 0: k-steps 0..41, class 1: 42..69, class 2: 70..97.
// =========================================================================================
__global__ void conv_pack_dgrad_kernel(const float *w, __bf16 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KSTEPS * 64 * 8) return;
    const int jj = i & 7, l = (i >> 3) & 63, s = i >> 9;
    const int r = s < 42 ? 0 : (s < 70 ? 1 : 2);
    const int sl = s - (r == 0 ? 0 : (r == 1 ? 42 : 70));           // k-step inside the class
    const int j = sl / 14, rem = sl % 14, kf = rem >> 1, cp = rem & 1;
    const int co = cp * 16 + 8 * (l >> 5) + jj, ci = l & 31, kt = r + 3 * j;
    out[i] = (__bf16)w[((co * CH + ci) * KS + kt) * KS + kf];
}

struct ConvDgradParams {
    long long *stamps;
    const __bf16 *dy;
    const __bf16 *wpack;
    __bf16 *dx;
    int B, H, W, Ho, Wo, Rq;      // Rq rows q per workgroup and class (Rq * W <= 192)
    int pitch;                    // bytes per (padded) dy row in LDS (pick_row_pitch)
    int flip;                     // second half of the grid takes the wave roles in reverse
};

// M-tiles (32 pixels each) per wave and class of the input-gradient kernel, and how many A
// fragments are read ahead of their MFMA
constexpr int DG_NT = 6, DG_PIX = DG_NT * 32, DG_AD = 2;

// NS k-steps starting at class-local step S0 (class row count J = 3 for r = 0, else 2).
// The operands are SWAPPED (weights as the MFMA's A, dy pixels as its B): the product comes out
// transposed, each lane holding 4 consecutive ci of one pixel per register quad — 8-byte LDS
// writes in the epilogue instead of 2-byte ones.  Reads run DG_AD MFMAs ahead, order pinned.
template <int NS, int S0>
__device__ __forceinline__ void conv_dgrad_part(const char *img, const unsigned (&pixbase)[DG_NT], int pitch,
                                                const bf16x8 (&bf)[28], f32x16 (&acc)[DG_NT]) {
    constexpr int N = NS * DG_NT;
    auto frag = [&](int idx) {
        const int s = idx / DG_NT, i = idx % DG_NT;
        const int sl = S0 + s;
        const int j = sl / 14, rem = sl % 14, kf = rem >> 1, cp = rem & 1;
        const unsigned off = (unsigned)((2 - j) * pitch + (6 - kf) * PIX + cp * 32);
        return *reinterpret_cast<const bf16x8 *>(img + pixbase[i] + off);
    };
    bf16x8 a[DG_AD];
#pragma unroll
    for (int d = 0; d < DG_AD; ++d) a[d] = frag(d);
#pragma unroll
    for (int idx = 0; idx < N; ++idx) {
        acc[idx % DG_NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[idx / DG_NT], a[idx % DG_AD],
                                                                   idx < DG_NT ? f32x16{} : acc[idx % DG_NT], 0, 0, 0);
        if (idx + DG_AD < N) a[idx % DG_AD] = frag(idx + DG_AD);
    }
#pragma unroll
    for (int d = 0; d < DG_AD; ++d) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int idx = 0; idx < N; ++idx) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (idx + DG_AD < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

// Per item (Rq rows q of the three classes of one utterance) a workgroup spends 168 MFMAs on
// its longest waves — 5.4k cycles — so everything else in the item loop is written for
// instruction count and for few dependent LDS round trips (stamped: with runtime divisions in
// the staging and copy-out loops, and one read -> wait -> store per copied chunk, those cost
// 15k cycles per item, three times the MFMA loop; an LDS round trip beside a workgroup in its
// MFMA loop takes ~230 cycles, a VALU instruction ~11):
//   * roles 0 and 1 (class 0, 21 k-steps each — 7 weight-fragment registers to spare) own the
//     staging: the item-invariant part of each of their 14 DMA slots' offsets sits in those
//     spare registers; per item and slot they add the item's row offset and issue the DMA;
//   * rows above / below the image and the 6 + 6 padding pixels are out-of-range reads (zeros):
//     the buffer descriptor spans ONE utterance, offsets below 0 wrap past it;
//   * roles 0 and 1 each finish half of class 0 (three tiles: the other's partial sums come
//     through LDS, all twelve reads in one round trip); every wave converts, transposes and
//     copies out its OWN pixels, reads batched ahead of the stores: two barriers per item.
// role 0: class 0, k-steps 0..20; role 1: class 0, 21..41; role 2: class 1; role 3: class 2.
// One instantiation per role, entered once per wave: a role that is a run-time value inside one
// body costs register copies and spills wherever its paths meet.
template <int ROLE>
__device__ __forceinline__ void conv_dgrad_wave(const ConvDgradParams &p, char *smem, int lane, int wave) {
    constexpr int NS = ROLE < 2 ? 21 : 28;                      // k-steps
    constexpr int FIRST = ROLE == 0 ? 0 : (ROLE == 1 ? 21 : (ROLE == 2 ? 42 : 70));
    constexpr int R = ROLE < 2 ? 0 : ROLE - 1;                  // class
    constexpr int HT = DG_NT / 2;
    constexpr int NT = ROLE < 2 ? HT : DG_NT, T0 = ROLE == 1 ? HT : 0;       // tiles finished here
    constexpr int GIVE0 = ROLE == 0 ? HT : 0;                                  // tiles given away (roles 0, 1)
    constexpr int SLOTS = 14;                                   // DMA slots of roles 0 and 1 each
    const int W = p.W, Wo = p.Wo, Rq = p.Rq;
    const unsigned img_bytes = (unsigned)(Rq + 2) * (unsigned)p.pitch;
    // ---- weight fragments: loaded ONCE per workgroup (persistent grid, as in the forward
    // kernel: one item per workgroup re-read 100 KB of fragments for 9 KB of dy) ----------------
    bf16x8 bf[28];
    {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p.wpack) + (size_t)FIRST * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) bf[s] = src[(size_t)s * 64];
    }
    // DMA slot k = SLOTS * ROLE + kk = LDS bytes [1024 k + 16 lane, +16) of the linear image (rows
    // `pitch` apart, pixels 80 bytes apart, 6 padding pixels on both sides of a row): offset of
    // its dy bytes from the item's first staged row, or a value no item offset brings into range
    unsigned tab[SLOTS];
    if (ROLE < 2) {
#pragma unroll
        for (int kk = 0; kk < SLOTS; ++kk) {
            const unsigned beta = (unsigned)(SLOTS * ROLE + kk) * 1024u + (unsigned)lane * 16u;
            const unsigned row = beta / (unsigned)p.pitch, rem = beta - row * (unsigned)p.pitch;
            const unsigned col = (rem * 52429u) >> 22, within = rem - col * 80u;        // rem / 80 for rem < 2^16
            const int wo = (int)col - 6;
            const bool ok = beta < img_bytes && within < 64u && wo >= 0 && wo < Wo;
            tab[kk] = ok ? (row * (unsigned)Wo + (unsigned)wo) * 64u + within : 0xC0000000u;
        }
    }
    unsigned pixbase[DG_NT];
    const int npix = Rq * W;
#pragma unroll
    for (int i = 0; i < DG_NT; ++i) {
        int m = 32 * i + (lane & 31);
        if (m >= npix) m = npix - 1;
        const int q = m / W, w = m - q * W;
        pixbase[i] = (unsigned)(q * p.pitch + w * PIX + (lane >> 5) * 16);
    }
    const int nq = (p.H + 2) / 3;                               // q = 0 .. ceil(H / 3) - 1
    const int tiles = (nq + Rq - 1) / Rq, nitems = tiles * p.B;
    const int nslots = (int)((img_bytes + 1023u) / 1024u);
    // LDS behind the image: [A 12 KB | B 12 KB | class 1 | class 2].  A = role 0's partial sums
    // of tiles 3..5, then role 1's half of the class-0 image (pixels 96..191, in A's first 6 KB);
    // B = role 1's partial sums of tiles 0..2, then role 0's half (pixels 0..95).
    constexpr int HALF = HT * 4 * 64 * 16;
    char *epi = smem + nslots * 1024;
    char *part_out = epi + (ROLE == 0 ? 0 : HALF);              // roles 0, 1: the half given away
    char *part_in = epi + (ROLE == 0 ? HALF : 0);               //             the half received
    // `ocls` = where pixel 0 of this wave's class would be (role 1 holds pixels 96.. only)
    char *ocls = ROLE == 0 ? epi + HALF : ROLE == 1 ? epi - T0 * 2048 : epi + 2 * HALF + (R - 1) * DG_PIX * 64;
    const unsigned img_lds = conv_lds_addr(smem);
    const unsigned item_bytes = (unsigned)(p.Ho * Wo * 64);
    auto stage = [&](int it) {          // roles 0 and 1
        const int b = it / tiles, q0 = (it - b * tiles) * Rq;
        const rsrc_words yR = conv_raw_rsrc(reinterpret_cast<const char *>(p.dy) + (size_t)b * item_bytes, item_bytes);
        const unsigned base = (unsigned)((q0 - 2) * Wo * 64);
#pragma unroll
        for (int kk = 0; kk < SLOTS; ++kk)
            if (SLOTS * ROLE + kk < nslots)
                conv_dma16(yR, (unsigned)__builtin_amdgcn_readfirstlane((int)(img_lds + (SLOTS * ROLE + kk) * 1024u)),
                           tab[kk] + base, 0u);
    };
    if (ROLE < 2 && (int)blockIdx.x < nitems) stage(blockIdx.x);
    // epilogue constants: a wave's pixels are written 8 bytes per lane (pixel = lane & 31 of
    // tile i, ci quad g, half lane >> 5) with the 16-byte chunks of a pixel XORed by
    // (pixel >> 2) & 3 — the 32 pixels of a write otherwise sit on 4 bank groups — and read back
    // 16 bytes per lane in pixel order
    const unsigned wr_base = (unsigned)((lane & 31) * 64 + ((lane >> 5) << 3));
    const unsigned wr_swz = (unsigned)((lane >> 2) & 3);
    const unsigned qmagic = 65536u / (unsigned)W + 1u;          // m / W = (m * qmagic) >> 16 for m < 2^12
    const int cls_chunks = Rq * W * 4;
    int nst = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++nst) {
        const int b = item / tiles, q0 = (item - b * tiles) * Rq;
        STAMP(0);
        if (ROLE < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                // image landed; the previous item's epilogue is done with LDS
        f32x16 acc[DG_NT];              // (the first k-step starts them from a literal zero)
        STAMP(1);
        conv_dgrad_part<NS, ROLE == 1 ? 21 : 0>(smem, pixbase, p.pitch, bf, acc);
        STAMP(2);
        if (ROLE < 2) {
#pragma unroll
            for (int i = 0; i < HT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16 &c = acc[GIVE0 + i];
                    *reinterpret_cast<f32x4 *>(part_out + ((i * 4 + g) * 64 + lane) * 16) =
                        f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                }
        }
        STAMP(3);
        __syncthreads();                // image no longer read; the partial sums are in LDS
        STAMP(4);
        if (ROLE < 2 && item + (int)gridDim.x < nitems) stage(item + gridDim.x);
        // ---- convert tiles [T0, T0 + NT) -> this wave's pixels in LDS -----------------------------
        {
            constexpr int E = NT * 4, RD = ROLE < 2 ? E / 2 : E;        // entries = (tile, ci quad)
#pragma unroll
            for (int e0 = 0; e0 < E; e0 += RD) {
                f32x4 o[ROLE < 2 ? RD : 1];
                if (ROLE < 2) {         // (a round's reads all before its writes: they share LDS,
                                        //  and a write never reaches past the entries read so far)
#pragma unroll
                    for (int k = 0; k < RD; ++k)
                        o[k] = *reinterpret_cast<const f32x4 *>(part_in + ((e0 + k) * 64 + lane) * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int k = 0; k < RD; ++k) {
                    const int i = (e0 + k) / 4, g = (e0 + k) % 4;
                    const f32x16 &c = acc[T0 + i];
                    f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                    if (ROLE < 2) v += o[k];
                    bf16x4 o4 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4 *>(ocls + (T0 + i) * 2048 + wr_base + (((unsigned)g ^ wr_swz) << 4)) = o4;
                }
            }
        }
        STAMP(5);
        {
            // output rows h = 3 (q0 + q) + R, W x 64 bytes each; chunk e of the class image is bytes
            // [16 e, +16) of pixel m = e >> 2: 16 e bytes into row q's start + q rows of the other
            // classes.  Six reads (one LDS round trip), then their stores.
            char *xb = reinterpret_cast<char *>(p.dx) + ((size_t)b * p.H + 3 * q0 + R) * W * 64;
            const int qlim = (p.H - R + 2) / 3 - q0;                 // rows q < qlim exist
            int e0 = T0 * 128 + lane;
            asm volatile("" : "+v"(e0));        // (recompute the chunk addresses per item: hoisted, they spill)
#pragma unroll
            for (int rd = 0; rd < NT * 2; rd += 6) {
                u32x4 v[6];
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const unsigned e = (unsigned)(e0 + 64 * (rd + it)), m = e >> 2;
                    v[it] = *reinterpret_cast<const u32x4 *>(ocls + ((e ^ ((m >> 2) & 3u)) << 4));
                }
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const unsigned e = (unsigned)(e0 + 64 * (rd + it)), m = e >> 2, q = (m * qmagic) >> 16;
                    if ((int)e < cls_chunks && (int)q < qlim)
                        *reinterpret_cast<u32x4 *>(xb + (size_t)(e * 16u + q * (unsigned)(2 * W * 64))) = v[it];
                }
            }
        }
        STAMP(6); STAMP(7);
    }
}

// Per item (Rq rows q of the three classes of one utterance) a workgroup spends 168 MFMAs on
// its longest waves — 5.4k cycles — so everything else in the item loop is written for
// instruction count and for few dependent LDS round trips (stamped: with runtime divisions in
// the staging and copy-out loops, and one read -> wait -> store per copied chunk, those cost
// 15k cycles per item, three times the MFMA loop; an LDS round trip beside a workgroup in its
// MFMA loop takes ~230 cycles, a VALU instruction ~11):
//   * roles 0 and 1 (class 0, 21 k-steps each: registers to spare) own the staging: the
//     item-invariant part of each of their 14 DMA slots' offsets sits in registers; per item
//     and slot they add the item's row offset and issue the DMA;
//   * rows above / below the image and the 6 + 6 padding pixels are out-of-range reads (zeros):
//     the buffer descriptor spans ONE utterance, offsets below 0 wrap past it;
//   * roles 0 and 1 each finish half of class 0 (three tiles: the other's partial sums come
//     through LDS, six reads per round trip); every wave converts, transposes and copies out
//     its OWN pixels, reads batched ahead of the stores: two barriers per item.
// Roles 2 and 3 carry 28 k-steps against 21: the second half of the grid (the workgroups that
// share a CU with the first half's) takes the roles in reverse wave order, so that every SIMD
// gets one long and one short wave.
__global__ __launch_bounds__(256, 2) void conv7x7c32_dgrad_s3_kernel(ConvDgradParams p) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = __builtin_amdgcn_readfirstlane(p.flip && blockIdx.x >= (gridDim.x >> 1) ? 3 - wave : wave);
    if (role == 0) conv_dgrad_wave<0>(p, smem, lane, wave);
    else if (role == 1) conv_dgrad_wave<1>(p, smem, lane, wave);
    else if (role == 2) conv_dgrad_wave<2>(p, smem, lane, wave);
    else conv_dgrad_wave<3>(p, smem, lane, wave);
}


// =========================================================================================
// weight gradient, stride (3, 1):
//   dw[co, ci, kt, kf] = sum_{b, ho, wo} dy[b, ho, wo, co] * x[b, 3 ho + kt, wo + kf, ci]
// One 32 x 32 (co x ci) MFMA tile per tap, summed over pixels: the reduction index k of the
// MFMA is the PIXEL, which is the slow index of both images in memory — both operands come
// out of LDS through the transposing read ds_read_b64_tr_b16 (4 pixels x 16 channels per
// 16-lane group, delivered channel-major).  A k-step = 16 consecutive wo of one output row
// (rows padded with zero dy pixels up to a multiple of 16; the x pixels read beside them are
// whatever follows in the image: finite, times zero).  Persistent workgroups walk chunks of
// 8 output rows; wave w (of 7) accumulates tap row kt = w x all 7 kf in registers for its
// whole life; partial sums per workgroup, then one reduction kernel.
// =========================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const char *addr) {
    // two transposed 4 x 16 blocks: pixels k0..k0+3 and k0+4..k0+7 of the lane's k-half
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr + 4 * PIX));
    s16x8 v;
    v.s0 = lo.x; v.s1 = lo.y; v.s2 = lo.z; v.s3 = lo.w;
    v.s4 = hi.x; v.s5 = hi.y; v.s6 = hi.z; v.s7 = hi.w;
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int WGRAD_WGS = 256;    // persistent workgroups (one per CU: 7 waves each)
constexpr int WGRAD_NT = 448;
constexpr int NX_MAX = 8, ND_MAX = 3;          // 16-byte chunks a thread stages per chunk of rows

struct ConvWgradParams {
    const __bf16 *x, *dy;
    float *partial;               // [workgroups][49][16][64]
    int B, H, W, Ho, Wo, KW;      // KW = ceil(Wo / 16) k-steps per output row
    int rows;                     // output rows per chunk
};

__global__ __launch_bounds__(WGRAD_NT) void conv7x7c32_wgrad_s3_kernel(ConvWgradParams p) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, Wo = p.Wo, KW = p.KW, Wd = 16 * KW, rows = p.rows;
    const int xrows = 3 * (rows - 1) + KS;
    char *ximg = smem;                                        // [xrows][W] pixels (+ tail pad)
    char *dimg = smem + (size_t)(xrows * W + 24) * PIX;       // [rows][Wd] pixels
    const int kt = wave;                                      // 7 waves: one tap row each
    f32x16 acc[KS];
#pragma unroll
    for (int f = 0; f < KS; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[f][j] = 0.f;
    // lane part of every transposed-read address: pixel 8 h + q, channels 16 chalf + 4 pp
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * PIX +
                                         (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    for (int i = tid; i < 24 * PIX / 4; i += WGRAD_NT)        // tail pad: finite values
        reinterpret_cast<unsigned *>(ximg + (size_t)xrows * W * PIX)[i] = 0u;
    const int chunks_per_utt = (p.Ho + rows - 1) / rows;
    const int nchunks = p.B * chunks_per_utt;
    const int nx = xrows * W * 4, nd = rows * Wd * 4;
    u32x4 sx[NX_MAX], sd[ND_MAX];
    // pix / W, pix / Wd by multiplication (pix < 2^12, divisors <= 64: exact)
    const unsigned wmagic = (1u << 20) / (unsigned)W + 1u, dmagic = (1u << 20) / (unsigned)Wd + 1u;
    // global -> registers: x rows 3 ho0 .. 3 ho0 + xrows - 1 (zeros past H), dy rows ho0 .. (zero
    // pixels up to Wd and past Ho)
    auto fetch = [&](int c) {
        const int b = c / chunks_per_utt, ho0 = (c - b * chunks_per_utt) * rows;
        const char *xb = reinterpret_cast<const char *>(p.x) + (size_t)b * p.H * W * 64;
        const char *yb = reinterpret_cast<const char *>(p.dy) + (size_t)b * p.Ho * Wo * 64;
#pragma unroll
        for (int k = 0; k < NX_MAX; ++k) {
            const int i = tid + k * WGRAD_NT;
            const int pix = i >> 2, part = i & 3, row = (int)(((unsigned)pix * wmagic) >> 20);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < nx && 3 * ho0 + row < p.H)
                v = *reinterpret_cast<const u32x4 *>(xb + ((size_t)(3 * ho0) * W + pix) * 64 + part * 16);
            sx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < ND_MAX; ++k) {
            const int i = tid + k * WGRAD_NT;
            const int pix = i >> 2, part = i & 3, row = (int)(((unsigned)pix * dmagic) >> 20), wo = pix - row * Wd;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < nd && ho0 + row < p.Ho && wo < Wo)
                v = *reinterpret_cast<const u32x4 *>(yb + ((size_t)(ho0 + row) * Wo + wo) * 64 + part * 16);
            sd[k] = v;
        }
    };
    auto stage = [&]() {                                      // registers -> LDS images
#pragma unroll
        for (int k = 0; k < NX_MAX; ++k) {
            const int i = tid + k * WGRAD_NT;
            if (i < nx) *reinterpret_cast<u32x4 *>(ximg + (i >> 2) * PIX + (i & 3) * 16) = sx[k];
        }
#pragma unroll
        for (int k = 0; k < ND_MAX; ++k) {
            const int i = tid + k * WGRAD_NT;
            if (i < nd) *reinterpret_cast<u32x4 *>(dimg + (i >> 2) * PIX + (i & 3) * 16) = sd[k];
        }
    };
    int c = blockIdx.x;
    if (c < nchunks) fetch(c);
    __syncthreads();
    if (c < nchunks) stage();
    __syncthreads();
    for (; c < nchunks; c += gridDim.x) {
        const int next = c + gridDim.x;
        if (next < nchunks) fetch(next);                      // in flight under this chunk's MFMAs
        for (int r = 0; r < rows; ++r) {
            for (int ks = 0; ks < KW; ++ks) {
                const bf16x8 a = tr_frag(dimg + (size_t)(r * Wd + 16 * ks) * PIX + lane_off);
                const char *xr = ximg + (size_t)((3 * r + kt) * W + 16 * ks) * PIX + lane_off;
#pragma unroll
                for (int f = 0; f < KS; ++f) {
                    const bf16x8 bb = tr_frag(xr + f * PIX);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[f], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                      // every wave is done with the images
        if (next < nchunks) stage();
        __syncthreads();
    }
    // partial sums, raw accumulator order [tap][reg][lane]
    float *out = p.partial + (size_t)blockIdx.x * 49 * 1024;
#pragma unroll
    for (int f = 0; f < KS; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) out[((kt * KS + f) * 16 + j) * 64 + lane] = acc[f][j];
}

// sum of `nparts` partial images of `n` floats: 64 consecutive elements per 256-thread block,
// the partials split over the block's waves (coalesced 256-byte reads)
__device__ __forceinline__ float partial_sum(const float *partial, int nparts, int n, float *lds) {
    const int el = threadIdx.x & 63, part = threadIdx.x >> 6, ways = blockDim.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < n)
        for (int g = part; g < nparts; g += ways) s += partial[(size_t)g * n + e];
    lds[part * 64 + el] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < ways; ++w) t += lds[w * 64 + el];
    return t;
}

// dw[co][ci][kt][kf] = sum over workgroups of partial[wg][tap][reg][lane]
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float *partial, int nwg, float *dw) {
    __shared__ float lds4[1024];
    const float s = partial_sum(partial, nwg, 49 * 1024, lds4);
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    if (threadIdx.x >= 64 || e >= 49 * 1024) return;
    const int tap = e >> 10, j = (e >> 6) & 15, l = e & 63;
    const int co = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), ci = l & 31;
    dw[(co * CH + ci) * 49 + tap] = s;
}


// =========================================================================================
// First convolution of the front-end: Conv2d(1, 32, (7, 7), stride (1, 2), padding (6, 0))
// (reference deep_speech_2.py:52-66) on the raw features x [B, T, F] fp32 ->
// y [B, To = T + 6, Fo = (F - 7) / 2 + 1, 32] bf16 channels-last.  49 taps padded to K = 64:
// k = kt * 8 + kf (kf = 7 and kt = 7 carry zero weights).  A workgroup turns the input rows of
// 16 output rows into a WINDOW image in LDS — for every (row, fo) the 8 consecutive samples
// x[row][2 fo .. 2 fo + 7] as bf16, 16 bytes — so an MFMA A-fragment (one tap row of one
// pixel) is one aligned ds_read_b128 and, for the weight gradient, a 4-tap group of one pixel
// is one aligned 8-byte piece for ds_read_b64_tr_b16.  HBM-bound: the 32-channel bf16 output
// (forward) / gradient (weight gradient) dominates, the MFMA work is 4 k-steps per 32 pixels.
// =========================================================================================
constexpr int C1_ROWS = 16;       // output rows per chunk

__global__ void conv1_pack_kernel(const float *w, __bf16 *out) {      // [4 k-steps][64][8]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 64 * 8) return;
    const int j = i & 7, l = (i >> 3) & 63, s = i >> 9;
    const int kt = 2 * s + (l >> 5), kf = j, co = l & 31;
    out[i] = (kt < KS && kf < KS) ? (__bf16)w[(co * KS + kt) * KS + kf] : (__bf16)0.f;
}

struct Conv1Params {
    const float *x;               // [B, T, F]
    const __bf16 *wpack;
    __bf16 *y;                    // forward: out; weight gradient: dy
    float *partial;
    int B, T, F, To, Fo;
};

// window image of input rows t0-6 .. t0-6+nrows-1: [nrows][Fo] x 8 bf16 (a window's 8 samples
// are four 8-byte loads: 2 fo + 7 <= F - 1 by the definition of Fo; the row index by
// multiplication: i / Fo = (i * fmagic) >> 20 for i * Fo < 2^20)
__device__ __forceinline__ void conv1_stage_windows(const Conv1Params &p, int b, int t0, int nrows,
                                                    char *wimg, int tid, int nt) {
    const float *xb = p.x + (size_t)b * p.T * p.F;
    const unsigned fmagic = (1u << 20) / (unsigned)p.Fo + 1u;
    for (int i = tid; i < nrows * p.Fo; i += nt) {
        const unsigned row = ((unsigned)i * fmagic) >> 20, fo = (unsigned)i - row * (unsigned)p.Fo;
        const int t = t0 - 6 + (int)row;
        bf16x8 v = {};
        if (t >= 0 && t < p.T) {
            const float2 *src = reinterpret_cast<const float2 *>(xb + (size_t)t * p.F + 2 * fo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 f2 = src[j];
                v[2 * j] = (__bf16)f2.x;
                v[2 * j + 1] = (__bf16)f2.y;
            }
        }
        *reinterpret_cast<bf16x8 *>(wimg + (size_t)i * 16) = v;
    }
}

__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1Params p) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, t0 = blockIdx.x * C1_ROWS, Fo = p.Fo;
    const int npix = C1_ROWS * Fo, ntiles = (npix + 31) >> 5;
    char *wimg = smem;                                         // [C1_ROWS + 8][Fo] windows
    __bf16 *oimg = reinterpret_cast<__bf16 *>(smem + (size_t)(C1_ROWS + 8) * Fo * 16);   // [ntiles*32][32]
    bf16x8 bf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bf[s] = reinterpret_cast<const bf16x8 *>(p.wpack)[s * 64 + lane];
    conv1_stage_windows(p, b, t0, C1_ROWS + 8, wimg, tid, 256);   // rows past the 7th tap: finite
    __syncthreads();
    const unsigned fmagic = (1u << 20) / (unsigned)Fo + 1u;
    for (int tile = wave; tile < ntiles; tile += 4) {
        int m = 32 * tile + (lane & 31);
        if (m >= npix) m = npix - 1;
        const unsigned r = ((unsigned)m * fmagic) >> 20, fo = (unsigned)m - r * (unsigned)Fo;
        const char *base = wimg + (size_t)((r + (unsigned)(lane >> 5)) * (unsigned)Fo + fo) * 16;
        bf16x8 a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const bf16x8 *>(base + (size_t)(2 * s * Fo) * 16);
        // operands swapped (weights as the MFMA's A): a lane ends up with 4 consecutive co of
        // pixel lane & 31 per register quad — 8-byte LDS writes instead of 2-byte ones
        f32x16 acc = {};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s], a[s], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o4 = {(__bf16)acc[4 * g], (__bf16)acc[4 * g + 1], (__bf16)acc[4 * g + 2], (__bf16)acc[4 * g + 3]};
            *reinterpret_cast<bf16x4 *>(reinterpret_cast<char *>(oimg) + (size_t)(32 * tile + (lane & 31)) * 64 +
                                        16 * g + 8 * (lane >> 5)) = o4;
        }
    }
    __syncthreads();
    const int rows_here = (p.To - t0) < C1_ROWS ? (p.To - t0) : C1_ROWS;
    char *yb = reinterpret_cast<char *>(p.y) + ((size_t)b * p.To + t0) * Fo * 64;
    for (int c = tid; c < rows_here * Fo * 4; c += 256)
        *reinterpret_cast<u32x4 *>(yb + (size_t)c * 16) =
            *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(oimg) + (size_t)c * 16);
    if (p.partial)      // channel sums of the outputs; scratch behind the output image
        chan_partial_sums(oimg, rows_here * Fo, reinterpret_cast<float *>(oimg + (size_t)ntiles * 32 * CH),
                          p.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64);
}

// weight gradient: dw[co][kt][kf] = sum_{b,t,fo} dy[b,t,fo,co] * x[b, t + kt - 6, 2 fo + kf]
// M = co, N = 64 taps (two tiles), K = the 16 * Fo pixels of a chunk in flattened order.
constexpr int C1_WGS = 1024;     // 4 workgroups per CU: the chunks' load latencies overlap
// DI = 16-byte pieces of a chunk's dy per thread, WPT = windows per thread (see the host).
// The NEXT chunk's dy pieces and samples are loaded into registers before the current chunk's
// MFMAs and stored to LDS behind them: the chunk loop then waits for HBM once per chunk at
// most, not once per staging loop (first version: load -> LDS store loops with run-time
// divisions, 13 us per chunk and workgroup for 17 KB of dy and 34 MFMAs).
template <int DI, int WPT>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(Conv1Params p) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Fo = p.Fo, npix = C1_ROWS * Fo, nks = npix >> 4;            // 16 | npix
    const int nwin = (C1_ROWS + 8) * Fo;
    char *wimg = smem;                                                    // [C1_ROWS + 8][Fo] windows
    char *dimg = smem + (size_t)nwin * 16;                                // [npix] pixels x 80 B
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const int h = lane >> 5, q = (lane & 15) >> 2, pp = lane & 3, chalf = (lane >> 4) & 1;
    const unsigned a_off = (unsigned)((8 * h + q) * PIX + (16 * chalf + 4 * pp) * 2);
    // B: taps 32 nt + 16 chalf + 4 pp .. + 3  ->  kt = 4 nt + 2 chalf + (pp >> 1), kf = 4 (pp & 1)
    const int ktl = 2 * chalf + (pp >> 1);
    const unsigned b_lane = (unsigned)(8 * (pp & 1));
    const int chunks_per_utt = (p.To + C1_ROWS - 1) / C1_ROWS;
    const int nchunks = p.B * chunks_per_utt;
    const unsigned fmagic = (1u << 20) / (unsigned)Fo + 1u;      // i / Fo = (i * fmagic) >> 20 for i * Fo < 2^20
    u32x4 dreg[DI];
    float2 wreg[WPT][4];
    auto fetch = [&](int c) {
        const int b = c / chunks_per_utt, t0 = (c - b * chunks_per_utt) * C1_ROWS;
        const char *yb = reinterpret_cast<const char *>(p.y) + ((size_t)b * p.To + t0) * Fo * 64;
        const int rows_here = (p.To - t0) < C1_ROWS ? (p.To - t0) : C1_ROWS;
#pragma unroll
        for (int k = 0; k < DI; ++k) {
            const int i = tid + 256 * k, pix = i >> 2;
            dreg[k] = pix < rows_here * Fo ? *reinterpret_cast<const u32x4 *>(yb + (size_t)i * 16) : u32x4{0u, 0u, 0u, 0u};
        }
        const float *xb = p.x + (size_t)b * p.T * p.F;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const unsigned i = (unsigned)(tid + 256 * u);
            const unsigned row = (i * fmagic) >> 20, fo = i - row * (unsigned)Fo;
            const int t = t0 - 6 + (int)row;
            const bool ok = (int)i < nwin && t >= 0 && t < p.T;
            const float2 *src = reinterpret_cast<const float2 *>(xb + (size_t)(ok ? t : 0) * p.F + 2 * fo);
#pragma unroll
            for (int j = 0; j < 4; ++j) wreg[u][j] = ok ? src[j] : float2{0.f, 0.f};
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < DI; ++k) {
            const int i = tid + 256 * k, pix = i >> 2, part = i & 3;
            if (pix < npix) *reinterpret_cast<u32x4 *>(dimg + pix * PIX + part * 16) = dreg[k];
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int i = tid + 256 * u;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = (__bf16)wreg[u][j].x; v[2 * j + 1] = (__bf16)wreg[u][j].y; }
            if (i < nwin) *reinterpret_cast<bf16x8 *>(wimg + (size_t)i * 16) = v;
        }
    };
    if ((int)blockIdx.x < nchunks) fetch(blockIdx.x);
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        __syncthreads();                // the previous chunk's MFMAs are done with LDS
        stash();
        __syncthreads();
        if (c + (int)gridDim.x < nchunks) fetch(c + gridDim.x);
        for (int ks = wave; ks < nks; ks += 4) {
            const bf16x8 a = tr_frag(dimg + (size_t)(16 * ks) * PIX + a_off);
            // window addresses of this lane's two pixel quartets
            const unsigned p0 = (unsigned)(16 * ks + 8 * h + q), p1 = p0 + 4;
            const unsigned r0 = (p0 * fmagic) >> 20, f0 = p0 - r0 * (unsigned)Fo;
            const unsigned r1 = (p1 * fmagic) >> 20, f1 = p1 - r1 * (unsigned)Fo;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int kt = 4 * nt + ktl;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4 *)(wimg + (size_t)((r0 + kt) * Fo + f0) * 16 + b_lane));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4 *)(wimg + (size_t)((r1 + kt) * Fo + f1) * 16 + b_lane));
                s16x8 v;
                v.s0 = lo.x; v.s1 = lo.y; v.s2 = lo.z; v.s3 = lo.w;
                v.s4 = hi.x; v.s5 = hi.y; v.s6 = hi.z; v.s7 = hi.w;
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, v), acc[nt], 0, 0, 0);
            }
        }
    }
    // the four waves' sums -> one partial image per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);                         // [4][2][16][64]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) red[((wave * 2 + nt) * 16 + j) * 64 + lane] = acc[nt][j];
    __syncthreads();
    float *out = p.partial + (size_t)blockIdx.x * 2 * 1024;
    for (int e = tid; e < 2 * 1024; e += 256)
        out[e] = red[e] + red[2048 + e] + red[4096 + e] + red[6144 + e];
}

__global__ __launch_bounds__(1024) void conv1_wgrad_reduce_kernel(const float *partial, int nparts, float *dw) {
    __shared__ float lds4[1024];
    const float s = partial_sum(partial, nparts, 2 * 1024, lds4);
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    if (threadIdx.x >= 64) return;
    const int nt = e >> 10, j = (e >> 6) & 15, l = e & 63;
    const int co = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), tap = 32 * nt + (l & 31);
    const int kt = tap >> 3, kf = tap & 7;
    if (kt < KS && kf < KS) dw[(co * KS + kt) * KS + kf] = s;
}


// =========================================================================================
// The same first convolution for CIN input channels in the reference's feature layout
// x [B, T, F, CIN] fp32 (deep_speech_2.py:127 permutes bs x t x f x c to NCHW; the WSJ recipes
// feed 81 mel bins x 3 channels — static, delta, delta-delta — egs/wsj/yamls/ctc.yaml:8-15):
// Conv2d(CIN, 32, (7, 7), stride (1, 2), padding (6, 0)).  K = CIN x 64: the single-channel
// scheme with one window image PER INPUT CHANNEL in LDS ([ci][row][fo] x 16 bytes) and 4 CIN
// k-steps per 32 pixels; the weight gradient has 2 CIN tap tiles.  ROWS output rows per
// chunk (8 for the 38-wide WSJ rows: 16 | ROWS * Fo, prefetch registers that fit).
// =========================================================================================
template <int CIN>
__global__ void conv1c_pack_kernel(const float *w, __bf16 *out) {      // [CIN][4 k-steps][64][8]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CIN * 4 * 64 * 8) return;
    const int j = i & 7, l = (i >> 3) & 63, s = (i >> 9) & 3, ci = i >> 11;
    const int kt = 2 * s + (l >> 5), kf = j, co = l & 31;
    out[i] = (kt < KS && kf < KS) ? (__bf16)w[((co * CIN + ci) * KS + kt) * KS + kf] : (__bf16)0.f;
}

// Staging.  The chunk's input rows t0-6 .. t0-6+nrows-1 are ONE contiguous piece of x
// (nrows * F * CIN floats): coalesced loads (RPT per thread, prefetchable into registers),
// de-interleaved into bf16 planes [ci][row][FP] in LDS (FP >= 2 (Fo-1) + 8 columns, the ones
// past F zero: for odd F the last window's 8th sample — zero weight — lies outside the row),
// from which the window image [ci][row][fo] x 8 samples is built LDS -> LDS.  (Gathering the
// windows straight from global memory — 8 strided loads per window — ran at 1.4 TB/s.)
template <int CIN, int RPT>
__device__ __forceinline__ void conv1c_raw_load(const Conv1Params &p, const float *xb, int t0, int nrows,
                                                int tid, float (&r)[RPT]) {
    const int FC = p.F * CIN;
    const unsigned mg = ((1u << 24) + (unsigned)FC - 1u) / (unsigned)FC;   // e / FC = (e * mg) >> 24 for e < 2^13 (host)
    const float *base = xb + (ptrdiff_t)(t0 - 6) * FC;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int e = tid + 256 * k;
        const int t = t0 - 6 + (int)(((unsigned)e * mg) >> 24);
        r[k] = (e < nrows * FC && t >= 0 && t < p.T) ? base[e] : 0.f;
    }
}
template <int CIN, int RPT>
__device__ __forceinline__ void conv1c_raw_stash(const Conv1Params &p, int nrows, int FP, int tid,
                                                 const float (&r)[RPT], __bf16 *planes) {
    const int FC = p.F * CIN;
    const unsigned mg = ((1u << 24) + (unsigned)FC - 1u) / (unsigned)FC;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int e = tid + 256 * k;
        if (e < nrows * FC) {
            const int row = (int)(((unsigned)e * mg) >> 24), rem = e - row * FC, f = rem / CIN, ci = rem - f * CIN;
            planes[(ci * nrows + row) * FP + f] = (__bf16)r[k];
        }
    }
    for (int i = tid; i < CIN * nrows * (FP - p.F); i += 256) {       // the pad columns
        const int rp = i / (FP - p.F), f = p.F + (i - rp * (FP - p.F));
        planes[rp * FP + f] = (__bf16)0.f;
    }
}
template <int CIN>
__device__ __forceinline__ void conv1c_build_windows(int nrows, int Fo, int FP, int tid,
                                                     const __bf16 *planes, char *wimg) {
    const unsigned fmagic = (1u << 20) / (unsigned)Fo + 1u;
    for (int i = tid; i < CIN * nrows * Fo; i += 256) {
        const unsigned rp = ((unsigned)i * fmagic) >> 20, fo = (unsigned)i - rp * (unsigned)Fo;   // rp = ci * nrows + row
        const unsigned *src = reinterpret_cast<const unsigned *>(planes + (size_t)rp * FP + 2 * fo);
        u32x4 v;
        v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
        *reinterpret_cast<u32x4 *>(wimg + (size_t)i * 16) = v;
    }
}

// Persistent: a workgroup walks chunks (utterance, ROWS output rows) with the NEXT chunk's input
// rows prefetched into registers while the current one is multiplied — one workgroup per chunk
// left every chunk's global round trip exposed (531 us at B = 256 x 1000 frames; with three
// workgroups per CU nothing else covers it).
template <int CIN, int ROWS, int RPT>
__global__ __launch_bounds__(256) void conv1c_fwd_kernel(Conv1Params p) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Fo = p.Fo;
    const int npix = ROWS * Fo, ntiles = (npix + 31) >> 5, nwin1 = (ROWS + 8) * Fo;
    const int FP = (2 * (Fo - 1) + 8 + 3) & ~3;
    char *wimg = smem;                                         // [CIN][ROWS + 8][Fo] windows
    __bf16 *oimg = reinterpret_cast<__bf16 *>(smem + (size_t)CIN * nwin1 * 16);   // [ntiles*32][32]
    __bf16 *planes = oimg;                                     // (dead before the first output is written)
    const unsigned fmagic = (1u << 20) / (unsigned)Fo + 1u;
    bf16x8 bw[CIN * 4];
#pragma unroll
    for (int k = 0; k < CIN * 4; ++k) bw[k] = reinterpret_cast<const bf16x8 *>(p.wpack)[k * 64 + lane];
    const int chunks_per_utt = (p.To + ROWS - 1) / ROWS;
    const int nchunks = p.B * chunks_per_utt;
    float rreg[RPT];
    auto fetch = [&](int c) {
        const int b = c / chunks_per_utt, t0 = (c - b * chunks_per_utt) * ROWS;
        conv1c_raw_load<CIN, RPT>(p, p.x + (size_t)b * p.T * p.F * CIN, t0, ROWS + 8, tid, rreg);
    };
    if ((int)blockIdx.x < nchunks) fetch(blockIdx.x);
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int b = c / chunks_per_utt, t0 = (c - b * chunks_per_utt) * ROWS;
        __syncthreads();                // the previous chunk's output image has left LDS
        conv1c_raw_stash<CIN, RPT>(p, ROWS + 8, FP, tid, rreg, planes);
        __syncthreads();
        if (c + (int)gridDim.x < nchunks) fetch(c + gridDim.x);
        conv1c_build_windows<CIN>(ROWS + 8, Fo, FP, tid, planes, wimg);
        __syncthreads();
        for (int tile = wave; tile < ntiles; tile += 4) {
            int m = 32 * tile + (lane & 31);
            if (m >= npix) m = npix - 1;
            const unsigned r = ((unsigned)m * fmagic) >> 20, fo = (unsigned)m - r * (unsigned)Fo;
            const char *base = wimg + (size_t)((r + (unsigned)(lane >> 5)) * (unsigned)Fo + fo) * 16;
            f32x16 acc = {};
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(base + (size_t)(ci * nwin1 + 2 * s4 * Fo) * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ci * 4 + s4], a, acc, 0, 0, 0);
                }
            // (tiles are written where the planes were: every wave is past the window build)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o4 = {(__bf16)acc[4 * g], (__bf16)acc[4 * g + 1], (__bf16)acc[4 * g + 2], (__bf16)acc[4 * g + 3]};
                *reinterpret_cast<bf16x4 *>(reinterpret_cast<char *>(oimg) + (size_t)(32 * tile + (lane & 31)) * 64 +
                                            16 * g + 8 * (lane >> 5)) = o4;
            }
        }
        __syncthreads();
        const int rows_here = (p.To - t0) < ROWS ? (p.To - t0) : ROWS;
        char *yb = reinterpret_cast<char *>(p.y) + ((size_t)b * p.To + t0) * Fo * 64;
        for (int cc = tid; cc < rows_here * Fo * 4; cc += 256)
            *reinterpret_cast<u32x4 *>(yb + (size_t)cc * 16) =
                *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(oimg) + (size_t)cc * 16);
        if (p.partial)
            chan_partial_sums(oimg, rows_here * Fo, reinterpret_cast<float *>(oimg + (size_t)ntiles * 32 * CH),
                              p.partial + (size_t)c * 64);
    }
}

// weight gradient: M = co, N = CIN x 64 taps (2 CIN tiles), K = the ROWS * Fo pixels of a chunk
template <int CIN, int ROWS, int DI, int RPT>
__global__ __launch_bounds__(256) void conv1c_wgrad_kernel(Conv1Params p) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Fo = p.Fo, npix = ROWS * Fo, nks = npix >> 4;               // 16 | npix (host)
    const int nwin1 = (ROWS + 8) * Fo;
    const int FP = (2 * (Fo - 1) + 8 + 3) & ~3;
    char *wimg = smem;                                                    // [CIN][ROWS + 8][Fo] windows
    char *dimg = smem + (size_t)CIN * nwin1 * 16;                         // [npix] pixels x 80 B
    __bf16 *planes = reinterpret_cast<__bf16 *>(dimg + (size_t)npix * PIX);   // [CIN][ROWS + 8][FP]
    f32x16 acc[2 * CIN];
#pragma unroll
    for (int t = 0; t < 2 * CIN; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const int h = lane >> 5, q = (lane & 15) >> 2, pp = lane & 3, chalf = (lane >> 4) & 1;
    const unsigned a_off = (unsigned)((8 * h + q) * PIX + (16 * chalf + 4 * pp) * 2);
    const int ktl = 2 * chalf + (pp >> 1);
    const unsigned b_lane = (unsigned)(8 * (pp & 1));
    const int chunks_per_utt = (p.To + ROWS - 1) / ROWS;
    const int nchunks = p.B * chunks_per_utt;
    const unsigned fmagic = (1u << 20) / (unsigned)Fo + 1u;
    u32x4 dreg[DI];
    float rreg[RPT];
    auto fetch = [&](int c) {
        const int b = c / chunks_per_utt, t0 = (c - b * chunks_per_utt) * ROWS;
        const char *yb = reinterpret_cast<const char *>(p.y) + ((size_t)b * p.To + t0) * Fo * 64;
        const int rows_here = (p.To - t0) < ROWS ? (p.To - t0) : ROWS;
#pragma unroll
        for (int k = 0; k < DI; ++k) {
            const int i = tid + 256 * k, pix = i >> 2;
            dreg[k] = pix < rows_here * Fo ? *reinterpret_cast<const u32x4 *>(yb + (size_t)i * 16) : u32x4{0u, 0u, 0u, 0u};
        }
        conv1c_raw_load<CIN, RPT>(p, p.x + (size_t)b * p.T * p.F * CIN, t0, ROWS + 8, tid, rreg);
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < DI; ++k) {
            const int i = tid + 256 * k, pix = i >> 2, part = i & 3;
            if (pix < npix) *reinterpret_cast<u32x4 *>(dimg + pix * PIX + part * 16) = dreg[k];
        }
        conv1c_raw_stash<CIN, RPT>(p, ROWS + 8, FP, tid, rreg, planes);
    };
    if ((int)blockIdx.x < nchunks) fetch(blockIdx.x);
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        __syncthreads();                // the previous chunk's MFMAs are done with LDS
        stash();
        __syncthreads();
        if (c + (int)gridDim.x < nchunks) fetch(c + gridDim.x);
        conv1c_build_windows<CIN>(ROWS + 8, Fo, FP, tid, planes, wimg);
        __syncthreads();
        for (int ks = wave; ks < nks; ks += 4) {
            const bf16x8 a = tr_frag(dimg + (size_t)(16 * ks) * PIX + a_off);
            const unsigned p0 = (unsigned)(16 * ks + 8 * h + q), p1 = p0 + 4;
            const unsigned r0 = (p0 * fmagic) >> 20, f0 = p0 - r0 * (unsigned)Fo;
            const unsigned r1 = (p1 * fmagic) >> 20, f1 = p1 - r1 * (unsigned)Fo;
#pragma unroll
            for (int nt = 0; nt < 2 * CIN; ++nt) {
                const int kt = 4 * (nt & 1) + ktl;
                const char *plane = wimg + (size_t)(nt >> 1) * nwin1 * 16;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4 *)(plane + (size_t)((r0 + kt) * Fo + f0) * 16 + b_lane));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4 *)(plane + (size_t)((r1 + kt) * Fo + f1) * 16 + b_lane));
                s16x8 v;
                v.s0 = lo.x; v.s1 = lo.y; v.s2 = lo.z; v.s3 = lo.w;
                v.s4 = hi.x; v.s5 = hi.y; v.s6 = hi.z; v.s7 = hi.w;
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, v), acc[nt], 0, 0, 0);
            }
        }
    }
    // the four waves' sums, one tap tile at a time -> one partial image per workgroup
    float *red = reinterpret_cast<float *>(smem);                         // [4][16][64]
    float *out = p.partial + (size_t)blockIdx.x * 2 * CIN * 1024;
#pragma unroll
    for (int nt = 0; nt < 2 * CIN; ++nt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) red[(wave * 16 + j) * 64 + lane] = acc[nt][j];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256)
            out[nt * 1024 + e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    }
}

template <int CIN>
__global__ __launch_bounds__(1024) void conv1c_wgrad_reduce_kernel(const float *partial, int nparts, float *dw) {
    __shared__ float lds4[1024];
    const float s = partial_sum(partial, nparts, 2 * CIN * 1024, lds4);
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    if (threadIdx.x >= 64) return;
    const int nt = e >> 10, j = (e >> 6) & 15, l = e & 63;
    const int ci = nt >> 1;
    const int co = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), tap = 32 * (nt & 1) + (l & 31);
    const int kt = tap >> 3, kf = tap & 7;
    if (kt < KS && kf < KS) dw[((co * CIN + ci) * KS + kt) * KS + kf] = s;
}

}  // namespace


#ifdef CONV_STAMPS
static long long *stamp_buf() {
    static long long *d = nullptr;
    if (!d && getenv("ASR_CONV_STAMPS")) { hipMalloc(&d, 4 * 64 * 8 * 8); }
    if (d) hipMemset(d, 0, 4 * 64 * 8 * 8);
    return d;
}
static hipEvent_t stamp_ev[2];
static void stamp_begin(hipStream_t s) {
    if (!stamp_ev[0]) { hipEventCreate(&stamp_ev[0]); hipEventCreate(&stamp_ev[1]); }
    hipEventRecord(stamp_ev[0], s);
}
static void stamp_dump(const char *name, long long *d, hipStream_t s) {
    if (!d) return;
    hipEventRecord(stamp_ev[1], s);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, stamp_ev[0], stamp_ev[1]);
    static long long h[4 * 64 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w)
        for (int lo = 2; lo < 30; lo += 14) {
            double avg[9] = {0};
            for (int it = lo; it < lo + 12; ++it) {
                for (int k = 1; k < 8; ++k) avg[k] += (double)(h[(w * 64 + it) * 8 + k] - h[(w * 64 + it) * 8 + k - 1]) / 12;
                avg[8] += (double)(h[(w * 64 + it + 1) * 8] - h[(w * 64 + it) * 8]) / 12;
            }
            fprintf(stderr, "%s wave %d items %d..%d avg:", name, w, lo, lo + 11);
            for (int k = 1; k < 8; ++k) fprintf(stderr, " %6.0f", avg[k]);
            fprintf(stderr, "  | total %.0f\n", avg[8]);
        }
    fprintf(stderr, "%s kernel %.1f us; items 0..30 of block 7 took %lld ticks\n", name, ms * 1e3, h[30 * 8] - h[0]);
}
#endif

extern "C" int64_t asr_conv7x7c32_workspace_bytes(void) {
    // two packed weight images (forward, input gradient) + the weight gradient's partial sums
    return (int64_t)KSTEPS * 64 * 8 * 2 * 2 + (int64_t)WGRAD_WGS * 49 * 1024 * 4 + 256;
}

extern "C" int asr_conv7x7c32_fwd_bf16(const void *x, const float *w, int B, int H, int W,
                                       int stride_h, void *y, double *chan_sums, void *workspace,
                                       int64_t workspace_bytes, void *stream) {
    if (B <= 0 || H < KS || W < KS) return ASR_EINVAL;
    if (!x || !w || !y || !workspace || workspace_bytes < asr_conv7x7c32_workspace_bytes())
        return ASR_EINVAL;
    if (stride_h != 1 && stride_h != 3) return ASR_EUNSUPPORTED;
    const int Ho = (H - KS) / stride_h + 1, Wo = W - KS + 1;
    if (Wo > 48) return ASR_EUNSUPPORTED;
    int R = FWD_PIX / Wo;
    if (R > 16) R = 16;
    const int pitch = pick_row_pitch(W * PIX, Wo, stride_h, R * Wo, 2 * FWD_NT);
    // (the image is filled in whole 1 KiB DMA blocks; behind it the slot table and the exchange area)
    const size_t img = ((size_t)(stride_h * (R - 1) + KS) * pitch + 1023) / 1024 * 1024;
    if ((int64_t)H * W * 64 >= (1ll << 31)) return ASR_EUNSUPPORTED;           // 32-bit buffer offsets
    if (img > (size_t)4 * FWD_MAXQ * 1024) return ASR_EUNSUPPORTED;
    const size_t lds = img + img / 8 + 4 * 4096 + 2048;
    if (lds > 80 * 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    __bf16 *wpack = (__bf16 *)workspace;
    hipLaunchKernelGGL(conv_pack_fwd_kernel, dim3((KSTEPS * 64 * 8 + 255) / 256), dim3(256), 0, s, w, wpack);
    ConvFwdParams p;
    p.x = (const __bf16 *)x; p.wpack = wpack; p.y = (__bf16 *)y;
    p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.R = R; p.pitch = pitch;
    const int nitems = ((Ho + R - 1) / R) * B;
    int wgs = 2 * conv_cu_count();
    if (wgs <= 0) wgs = 512;
    if (wgs > nitems) wgs = nitems;
    const dim3 grid(wgs);
    // the weight gradient's partial-sum area doubles as the statistics' (never live together)
    p.stats = chan_sums ? (float *)((char *)workspace + (size_t)KSTEPS * 64 * 8 * 2 * 2) : nullptr;
    if (chan_sums && (int64_t)wgs * 4 * 64 * 4 > (int64_t)WGRAD_WGS * 49 * 1024 * 4) return ASR_EUNSUPPORTED;
    void (*kern)(ConvFwdParams) = stride_h == 3 ? conv7x7c32_fwd_kernel<3> : conv7x7c32_fwd_kernel<1>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ASR_EUNSUPPORTED;
#ifdef CONV_STAMPS
    p.stamps = stamp_buf();
    stamp_begin(s);
#else
    p.stamps = nullptr;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
#ifdef CONV_STAMPS
    stamp_dump("fwd", p.stamps, s);
#endif
    if (chan_sums) {
        hipLaunchKernelGGL(zero_chan_sums_kernel, dim3(1), dim3(64), 0, s, chan_sums);
        hipLaunchKernelGGL(chan_sums_reduce_kernel, dim3(64), dim3(1024), 0, s, p.stats, wgs * 4, chan_sums);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_conv7x7c32_bwd_data_bf16(const void *dy, const float *w, int B, int H, int W,
                                            int stride_h, void *dx, void *workspace,
                                            int64_t workspace_bytes, void *stream) {
    if (B <= 0 || H < KS || W < KS) return ASR_EINVAL;
    if (!dy || !w || !dx || !workspace || workspace_bytes < asr_conv7x7c32_workspace_bytes())
        return ASR_EINVAL;
    if (stride_h != 3) return ASR_EUNSUPPORTED;
    const int Ho = (H - KS) / stride_h + 1, Wo = W - KS + 1;
    if (W > 48) return ASR_EUNSUPPORTED;
    int Rq = DG_PIX / W;
    if (Rq > 16) Rq = 16;
    const int pitch = pick_row_pitch((Wo + 12) * PIX, W, 1, Rq * W, DG_NT);
    // (the image is filled in whole 1 KiB DMA blocks)
    const size_t img = ((size_t)(Rq + 2) * pitch + 1023) / 1024 * 1024;
    if ((int64_t)B * Ho * Wo * 64 >= (1ll << 31)) return ASR_EUNSUPPORTED;     // 32-bit buffer offsets
    // epilogue beside the image: the partial-sum exchange (class 0 of the output image reuses
    // its first half) + classes 1 and 2
    const size_t epi = (size_t)DG_NT * 4 * 64 * 16 + (size_t)2 * DG_PIX * CH * 2;
    const size_t lds = img + epi;
    if (lds > 80 * 1024 || img > 28 * 1024) return ASR_EUNSUPPORTED;          // (28 DMA slots)
    hipStream_t s = (hipStream_t)stream;
    __bf16 *wpack = (__bf16 *)workspace + (size_t)KSTEPS * 64 * 8;
    hipLaunchKernelGGL(conv_pack_dgrad_kernel, dim3((KSTEPS * 64 * 8 + 255) / 256), dim3(256), 0, s, w, wpack);
    ConvDgradParams p;
    p.dy = (const __bf16 *)dy; p.wpack = wpack; p.dx = (__bf16 *)dx;
    p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Rq = Rq; p.pitch = pitch;
    const int nq = (H + 2) / 3;                                  // q = 0 .. ceil(H / 3) - 1
    const int nitems = ((nq + Rq - 1) / Rq) * B;
    int wgs = 2 * conv_cu_count();
    if (wgs <= 0) wgs = 512;
    p.flip = wgs <= nitems;                                      // (two workgroups on every CU)
    if (wgs > nitems) wgs = nitems;
    const dim3 grid(wgs);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)conv7x7c32_dgrad_s3_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ASR_EUNSUPPORTED;
#ifdef CONV_STAMPS
    p.stamps = stamp_buf();
    stamp_begin(s);
#else
    p.stamps = nullptr;
#endif
    hipLaunchKernelGGL(conv7x7c32_dgrad_s3_kernel, grid, dim3(256), lds, s, p);
#ifdef CONV_STAMPS
    stamp_dump("dgrad", p.stamps, s);
#endif
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_conv7x7c32_wgrad_bf16(const void *x, const void *dy, int B, int H, int W,
                                         int stride_h, float *dw, void *workspace,
                                         int64_t workspace_bytes, void *stream) {
    if (B <= 0 || H < KS || W < KS) return ASR_EINVAL;
    if (!x || !dy || !dw || !workspace || workspace_bytes < asr_conv7x7c32_workspace_bytes())
        return ASR_EINVAL;
    if (stride_h != 3) return ASR_EUNSUPPORTED;
    const int Ho = (H - KS) / stride_h + 1, Wo = W - KS + 1;
    if (Wo > 48) return ASR_EUNSUPPORTED;
    const int KW = (Wo + 15) / 16;
    int rows = 8;                                  // output rows per chunk: as many as LDS and
    size_t lds = 0;                                // the staging registers take
    for (; rows >= 1; rows >>= 1) {
        const int xrows = 3 * (rows - 1) + KS;
        lds = (size_t)(xrows * W + 24) * PIX + (size_t)rows * 16 * KW * PIX;
        if (lds <= 72 * 1024 && xrows * W * 4 <= NX_MAX * WGRAD_NT && rows * 16 * KW * 4 <= ND_MAX * WGRAD_NT)
            break;
    }
    if (rows < 1) return ASR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    ConvWgradParams p;
    p.x = (const __bf16 *)x; p.dy = (const __bf16 *)dy;
    p.partial = (float *)((char *)workspace + (size_t)KSTEPS * 64 * 8 * 2 * 2);
    p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KW = KW; p.rows = rows;
    const int chunks = B * ((Ho + rows - 1) / rows);
    const int nwg = chunks < WGRAD_WGS ? chunks : WGRAD_WGS;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)conv7x7c32_wgrad_s3_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ASR_EUNSUPPORTED;
    hipLaunchKernelGGL(conv7x7c32_wgrad_s3_kernel, dim3(nwg), dim3(WGRAD_NT), lds, s, p);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(49 * 1024 / 64), dim3(256), 0, s,
                       p.partial, nwg, dw);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int64_t asr_conv1_7x7s2_workspace_bytes(void) {
    // packed weights + max(weight-gradient partials, 64 channel sums per forward workgroup: up
    // to 2^17 workgroups)
    return 4 * 64 * 8 * 2 + (int64_t)(1 << 17) * 64 * 4 + 256;
}

static int conv1_shapes(int B, int T, int F, int *To, int *Fo) {
    if (B <= 0 || T <= 0 || F < KS) return ASR_EINVAL;
    *To = T + 2 * 6 - KS + 1;
    *Fo = (F - KS) / 2 + 1;
    if (*Fo > 64) return ASR_EUNSUPPORTED;
    return ASR_OK;
}

extern "C" int asr_conv1_7x7s2_fwd(const float *x, const float *w, int B, int T, int F, void *y,
                                   double *chan_sums, void *workspace, int64_t workspace_bytes,
                                   void *stream) {
    int To, Fo;
    const int rc = conv1_shapes(B, T, F, &To, &Fo);
    if (rc != ASR_OK) return rc;
    if (!x || !w || !y || !workspace || workspace_bytes < asr_conv1_7x7s2_workspace_bytes())
        return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    __bf16 *wpack = (__bf16 *)workspace;
    hipLaunchKernelGGL(conv1_pack_kernel, dim3(8), dim3(256), 0, s, w, wpack);
    Conv1Params p;
    p.x = x; p.wpack = wpack; p.y = (__bf16 *)y;
    p.B = B; p.T = T; p.F = F; p.To = To; p.Fo = Fo;
    const dim3 grid((To + C1_ROWS - 1) / C1_ROWS, B);
    // the weight gradient's partial-sum area doubles as the statistics' (never live together)
    p.partial = chan_sums ? (float *)((char *)workspace + 4 * 64 * 8 * 2) : nullptr;
    if (chan_sums && (int64_t)grid.x * grid.y * 64 * 4 > asr_conv1_7x7s2_workspace_bytes() - 4 * 64 * 8 * 2 - 256)
        return ASR_EUNSUPPORTED;
    const int ntiles = (C1_ROWS * Fo + 31) / 32;
    const size_t lds = (size_t)(C1_ROWS + 8) * Fo * 16 + (size_t)ntiles * 32 * CH * 2 + 512 * 4;
    if (lds > 64 * 1024) return ASR_EUNSUPPORTED;
    hipLaunchKernelGGL(conv1_fwd_kernel, grid, dim3(256), lds, s, p);
    if (chan_sums) {
        hipLaunchKernelGGL(zero_chan_sums_kernel, dim3(1), dim3(64), 0, s, chan_sums);
        hipLaunchKernelGGL(chan_sums_reduce_kernel, dim3(64), dim3(1024), 0, s, p.partial, (int)(grid.x * grid.y), chan_sums);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_conv1_7x7s2_wgrad(const float *x, const void *dy, int B, int T, int F,
                                     float *dw, void *workspace, int64_t workspace_bytes,
                                     void *stream) {
    int To, Fo;
    const int rc = conv1_shapes(B, T, F, &To, &Fo);
    if (rc != ASR_OK) return rc;
    if (!x || !dy || !dw || !workspace || workspace_bytes < asr_conv1_7x7s2_workspace_bytes())
        return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Conv1Params p;
    p.x = x; p.wpack = nullptr; p.y = (__bf16 *)const_cast<void *>(dy);
    p.partial = (float *)((char *)workspace + 4 * 64 * 8 * 2);
    p.B = B; p.T = T; p.F = F; p.To = To; p.Fo = Fo;
    size_t lds = (size_t)(C1_ROWS + 8) * Fo * 16 + (size_t)C1_ROWS * Fo * PIX;
    if (lds < 4 * 2 * 1024 * 4) lds = 4 * 2 * 1024 * 4;            // the final four-wave sum
    if (lds > 64 * 1024) return ASR_EUNSUPPORTED;
    const int chunks = B * ((To + C1_ROWS - 1) / C1_ROWS);
    const int nwg = chunks < C1_WGS ? chunks : C1_WGS;
    // pieces of dy / windows per thread (256 threads): template sizes of the prefetch registers
    const int di = (C1_ROWS * Fo * 4 + 255) / 256, wpt = ((C1_ROWS + 8) * Fo + 255) / 256;
    void (*kern)(Conv1Params) = di <= 5 && wpt <= 2 ? conv1_wgrad_kernel<5, 2>
                              : (di <= 8 && wpt <= 3 ? conv1_wgrad_kernel<8, 3> : conv1_wgrad_kernel<16, 6>);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, p);
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(2 * 1024 / 64), dim3(1024), 0, s, p.partial, nwg, dw);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}


// ---- first convolution with CIN input channels (x [B, T, F, CIN]) -------------------------
extern "C" int64_t asr_conv1c_7x7s2_workspace_bytes(int cin) {
    if (cin != 3) return -1;
    // packed weights + max(weight-gradient partials: 1024 workgroups x 2 cin tiles, channel sums)
    const int64_t wg = (int64_t)C1_WGS * 2 * cin * 1024 * 4, cs = (int64_t)(1 << 17) * 64 * 4;
    return (int64_t)cin * 4 * 64 * 8 * 2 + (wg > cs ? wg : cs) + 256;
}

extern "C" int asr_conv1c_7x7s2_fwd(const float *x, const float *w, int B, int T, int F, int cin, void *y,
                                    double *chan_sums, void *workspace, int64_t workspace_bytes,
                                    void *stream) {
    int To, Fo;
    const int rc = conv1_shapes(B, T, F, &To, &Fo);
    if (rc != ASR_OK) return rc;
    if (cin != 3 || (Fo & 1)) return ASR_EUNSUPPORTED;
    if (!x || !w || !y || !workspace || workspace_bytes < asr_conv1c_7x7s2_workspace_bytes(cin))
        return ASR_EINVAL;
    if ((int64_t)T * F * cin * 4 * B >= (1ll << 40)) return ASR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    __bf16 *wpack = (__bf16 *)workspace;
    hipLaunchKernelGGL(conv1c_pack_kernel<3>, dim3(3 * 8), dim3(256), 0, s, w, wpack);
    Conv1Params p;
    p.x = x; p.wpack = wpack; p.y = (__bf16 *)y;
    p.B = B; p.T = T; p.F = F; p.To = To; p.Fo = Fo;
    constexpr int ROWS = 8;
    const int64_t nchunks = (int64_t)B * ((To + ROWS - 1) / ROWS);
    if (nchunks >= (1ll << 31)) return ASR_EUNSUPPORTED;
    const dim3 grid((unsigned)(nchunks < 768 ? nchunks : 768));       // three workgroups per CU (LDS)
    p.partial = chan_sums ? (float *)((char *)workspace + 3 * 4 * 64 * 8 * 2) : nullptr;
    if (chan_sums && nchunks * 64 * 4 > (int64_t)(1 << 17) * 64 * 4) return ASR_EUNSUPPORTED;
    const int ntiles = (ROWS * Fo + 31) / 32;
    const int FP = (2 * (Fo - 1) + 8 + 3) & ~3;
    size_t otail = (size_t)ntiles * 32 * CH * 2 + 512 * 4;
    if (otail < (size_t)3 * (ROWS + 8) * FP * 2) otail = (size_t)3 * (ROWS + 8) * FP * 2;      // the planes alias it
    const size_t lds = (size_t)3 * (ROWS + 8) * Fo * 16 + otail;
    const int rpt = ((ROWS + 8) * F * 3 + 255) / 256;
    if (lds > 64 * 1024 || rpt > 24) return ASR_EUNSUPPORTED;
    if (rpt <= 16) hipLaunchKernelGGL((conv1c_fwd_kernel<3, ROWS, 16>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv1c_fwd_kernel<3, ROWS, 24>), grid, dim3(256), lds, s, p);
    if (chan_sums) {
        hipLaunchKernelGGL(zero_chan_sums_kernel, dim3(1), dim3(64), 0, s, chan_sums);
        hipLaunchKernelGGL(chan_sums_reduce_kernel, dim3(64), dim3(1024), 0, s, p.partial, (int)nchunks, chan_sums);
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

extern "C" int asr_conv1c_7x7s2_wgrad(const float *x, const void *dy, int B, int T, int F, int cin,
                                      float *dw, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
    int To, Fo;
    const int rc = conv1_shapes(B, T, F, &To, &Fo);
    if (rc != ASR_OK) return rc;
    if (cin != 3 || (Fo & 1)) return ASR_EUNSUPPORTED;
    if (!x || !dy || !dw || !workspace || workspace_bytes < asr_conv1c_7x7s2_workspace_bytes(cin))
        return ASR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    constexpr int ROWS = 8;
    Conv1Params p;
    p.x = x; p.wpack = nullptr; p.y = (__bf16 *)const_cast<void *>(dy);
    p.partial = (float *)((char *)workspace + 3 * 4 * 64 * 8 * 2);
    p.B = B; p.T = T; p.F = F; p.To = To; p.Fo = Fo;
    const int FP = (2 * (Fo - 1) + 8 + 3) & ~3;
    size_t lds = (size_t)3 * (ROWS + 8) * Fo * 16 + (size_t)ROWS * Fo * PIX + (size_t)3 * (ROWS + 8) * FP * 2;
    if (lds < 4 * 1024 * 4) lds = 4 * 1024 * 4;                    // the final four-wave sums
    if (lds > 64 * 1024) return ASR_EUNSUPPORTED;
    const int chunks = B * ((To + ROWS - 1) / ROWS);
    const int nwg = chunks < C1_WGS ? chunks : C1_WGS;
    const int di = (ROWS * Fo * 4 + 255) / 256, rpt = ((ROWS + 8) * F * 3 + 255) / 256;
    if (di > 8 || rpt > 24) return ASR_EUNSUPPORTED;
    void (*kern)(Conv1Params) = di <= 5 && rpt <= 16 ? conv1c_wgrad_kernel<3, ROWS, 5, 16>
                                                      : conv1c_wgrad_kernel<3, ROWS, 8, 24>;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, p);
    hipLaunchKernelGGL(conv1c_wgrad_reduce_kernel<3>, dim3(2 * 3 * 1024 / 64), dim3(1024), 0, s, p.partial, nwg, dw);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
