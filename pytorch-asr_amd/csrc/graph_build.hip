// Device-side construction of the CTC training lattices.
//
// Replaces, for context orders 1 and 2, the host pipeline of the reference:
// compose(decoding_fst, chain(labels)) with OpenFst in the data workers
// (att_speech/fst_utils.py:603-613, data/kaldi_dataset.py:230-232),
// fst_to_matrices (:222-294), batch_training_graph_matrices (:491-521) and the
// per-step copy of 8 padded tensors to the device
// (modules/decoders/advanced_decoder.py:457-459).  The composed lattice has a
// closed form (states 0 = initial blank, 2j+1 = label j, 2j+2 = blank after it;
// SURVEY.md §8a A4), so one thread per (utterance, state) writes that state's
// <= 3 incoming and <= 3 outgoing arcs straight into the int32 / f32 arrays the
// lattice kernels consume.  Arc order inside a state equals fst_to_matrices'
// sort by (other state, ilabel, weight).
#include "common.h"
#include "../../include/asr_amd.h"

namespace {

struct BuildParams {
    const int32_t *labels;     // [B,Lmax] symbols (already reduced modulo S)
    const int32_t *label_lens; // [B]
    int B, Lmax, S, order, allow_self, ctx_blank;
    float nc;
    int32_t *src_in, *il_in;   // [B,N,3]
    float *w_in, *term;        // [B,N,3], [B,N]
    int32_t *dst_out, *il_out; // [B,N,3]
    float *w_out;
};

__global__ void ctc_graph_build_kernel(BuildParams p) {
    const int N = 2 * p.Lmax + 1;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.B * N) return;
    const int b = (int)(idx / N), n = (int)(idx % N);
    int L = p.label_lens[b];
    L = L < 0 ? 0 : (L > p.Lmax ? p.Lmax : L);
    const int Nb = 2 * L + 1;
    const int32_t *lab = p.labels + (size_t)b * p.Lmax;
    const int S = p.S;
    auto sym = [&](int j) { return lab[j]; };
    auto ctx = [&](int j) { return j > 0 ? lab[j - 1] : 0; };
    auto emit = [&](int j) { return p.order == 1 ? sym(j) : ctx(j) * S + sym(j); };
    auto blank_after = [&](int j) {            // blank state that follows label j
        return (p.order == 2 && p.ctx_blank) ? sym(j) * S : 0;
    };
    auto selfloop = [&](int j) {
        return p.order == 1 ? true : (p.allow_self || ctx(j) == sym(j));
    };
    auto skip_ok = [&](int j) {                // label j -> label j+1 without a blank
        return p.order == 1 ? sym(j) != sym(j + 1)
                            : !(ctx(j) == sym(j) && sym(j) == sym(j + 1));
    };
    int si[3] = {0, 0, 0}, li[3] = {0, 0, 0}, so[3] = {0, 0, 0}, lo[3] = {0, 0, 0};
    int ni = 0, no = 0;
    if (n < Nb) {
        if (n == 0) {
            si[ni] = 0; li[ni++] = 0;                               // blank self-loop
            so[no] = 0; lo[no++] = 0;
            if (L >= 1) { so[no] = 1; lo[no++] = emit(0); }
        } else if (n & 1) {                                          // label j
            const int j = (n - 1) >> 1, e = emit(j);
            if (j >= 1 && skip_ok(j - 1)) { si[ni] = n - 2; li[ni++] = e; }
            si[ni] = n - 1; li[ni++] = e;                           // from the blank before
            if (selfloop(j)) { si[ni] = n; li[ni++] = e; }
            if (selfloop(j)) { so[no] = n; lo[no++] = e; }
            so[no] = n + 1; lo[no++] = blank_after(j);
            if (j + 1 < L && skip_ok(j)) { so[no] = n + 2; lo[no++] = emit(j + 1); }
        } else {                                                     // blank after label j
            const int j = (n >> 1) - 1, bl = blank_after(j);
            si[ni] = n - 1; li[ni++] = bl;
            si[ni] = n; li[ni++] = bl;
            so[no] = n; lo[no++] = bl;
            if (j + 1 < L) { so[no] = n + 1; lo[no++] = emit(j + 1); }
        }
    }
    const size_t o = ((size_t)b * N + n) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.src_in[o + k] = k < ni ? si[k] : 0;
        p.il_in[o + k] = k < ni ? li[k] : 0;
        p.w_in[o + k] = k < ni ? 0.f : p.nc;
        p.dst_out[o + k] = k < no ? so[k] : 0;
        p.il_out[o + k] = k < no ? lo[k] : 0;
        p.w_out[o + k] = k < no ? 0.f : p.nc;
    }
    const bool fin = n < Nb && (n == Nb - 1 || (L > 0 && n == Nb - 2));
    p.term[(size_t)b * N + n] = fin ? 0.f : p.nc;
}

}  // namespace

extern "C" int asr_ctc_graph_build(const int32_t *labels, const int32_t *label_lens, int B,
                                   int Lmax, int num_symbols, int context_order,
                                   int allow_nonblank_selfloops, int use_contextual_blanks,
                                   float nc_weight, int32_t *src_in, int32_t *il_in,
                                   float *w_in, float *term, int32_t *dst_out,
                                   int32_t *il_out, float *w_out, void *stream) {
    if (B < 0 || Lmax < 0 || num_symbols <= 0 || (context_order != 1 && context_order != 2))
        return ASR_EINVAL;
    if (B == 0) return ASR_OK;
    if (!label_lens || (Lmax > 0 && !labels) || !src_in || !il_in || !w_in || !term ||
        !dst_out || !il_out || !w_out || !(nc_weight < 0.f))
        return ASR_EINVAL;
    BuildParams p;
    p.labels = labels; p.label_lens = label_lens; p.B = B; p.Lmax = Lmax;
    p.S = num_symbols; p.order = context_order;
    p.allow_self = allow_nonblank_selfloops; p.ctx_blank = use_contextual_blanks;
    p.nc = nc_weight;
    p.src_in = src_in; p.il_in = il_in; p.w_in = w_in; p.term = term;
    p.dst_out = dst_out; p.il_out = il_out; p.w_out = w_out;
    const long long total = (long long)B * (2 * Lmax + 1);
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(ctc_graph_build_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
