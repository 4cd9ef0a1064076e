/*
 * include/asr_amd.h — C ABI of libasr_amd.so (MI355X / gfx950 HIP kernels).
 *
 * The reference (chorowski-lab/pytorch-asr) has no native code and no FFI for
 * this path: its lattice arithmetic is Python-level torch code resolved by name
 * (SURVEY.md §8b).  Each entry point below therefore replaces a *Python*
 * function of the reference; the citation names the file:line (relative to the
 * reference root) whose arithmetic the kernel reproduces.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_;
 *   - plain pointers and sizes only, no torch / HIP types (the stream is passed
 *     as void* = hipStream_t);
 *   - caller owns every buffer, the library never allocates, never
 *     synchronises, and launches only on the passed stream (graph-capturable);
 *   - return value: ASR_OK or an ASR_E* code; nothing is launched on error;
 *   - tensors are dense row-major; log-probs are TIME-MAJOR [T,B,C] like the
 *     reference (fst_utils.py:329);
 *   - graph matrices are the reference's padded adjacency form
 *     (fst_utils.py:222-294, 491-521) with int32 indices:
 *     [Bg,N,K] with Bg == 1 (shared, e.g. the denominator graph,
 *     fst_utils.py:662-676) or Bg == B; padding arcs carry weight <= neg_inf/2.
 */
#ifndef ASR_AMD_H
#define ASR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ASR_OK = 0,
    ASR_EINVAL = 1,      /* bad shape / null pointer (reference: AssertionError) */
    ASR_EUNSUPPORTED = 2,/* shape exceeds what the kernels are built for */
    ASR_ELAUNCH = 3      /* HIP reported a launch error */
};

/* Library / ABI version, bumped when a signature changes. */
int asr_abi_version(void);

/* Human-readable text for an ASR_* code (static storage). */
const char *asr_strerror(int code);

/*
 * Bytes of workspace asr_lattice_fwbw_f32 needs (alphas [T,B,N] f32 plus
 * internal scratch).  Replaces the `lalphas` allocation of
 * PathLogSumExp.forward (fst_utils.py:428).
 */
int64_t asr_lattice_fwbw_workspace_bytes(int T, int B, int C, int N);

/*
 * Log-semiring forward-backward over per-utterance lattices.
 * Replaces PathLogSumExp.forward (fst_utils.py:403-480), reached through
 * path_reduction(..., red_kind in {'logsumexp' with 8 matrices,
 * 'logsumexp_fwb'}) (fst_utils.py:345-347).
 *
 *   lp        [T,B,C] f32   log-probs, time-major
 *   lens      [B]     i32   active frames per utterance, 0 <= lens[b] <= T
 *                           (the reference additionally requires them sorted
 *                           descending, fst_utils.py:432; the kernel does not)
 *   src_in/il_in/w_in   [Bg,N,Kin]  incoming arcs of each state
 *   term                [Bg,N]      terminal log-weights
 *   dst_out/il_out/w_out [Bg,N,Kout] outgoing arcs of each state
 *   out_logZ  [B]     f32   +log-sum of all accepted paths (callers negate)
 *   out_grad  [T,B,C] f32   d logZ[b] / d lp[t,b,c]; rows t >= lens[b] are
 *                           written as zeros (fst_utils.py:448)
 *   out_logZ_bwd [B] f32 or NULL: the backward-pass total used by the
 *                           reference's consistency print (fst_utils.py:475-479)
 *   workspace: asr_lattice_fwbw_workspace_bytes(T,B,C,N) bytes
 */
int asr_lattice_fwbw_f32(const float *lp, int T, int B, int C,
                         const int32_t *lens,
                         const int32_t *src_in, const int32_t *il_in,
                         const float *w_in, const float *term,
                         const int32_t *dst_out, const int32_t *il_out,
                         const float *w_out,
                         int N, int Kin, int Kout, int Bg, float neg_inf,
                         float *out_logZ, float *out_grad,
                         float *out_logZ_bwd,
                         void *workspace, int64_t workspace_bytes,
                         void *stream);

/*
 * The same forward-backward (same arguments, outputs and workspace: replaces
 * PathLogSumExp.forward, fst_utils.py:403-480) for BAND lattices in the rescaled linear
 * domain (ABI v13; csrc/lattice_band.inc): state-labelled graphs whose state n is entered
 * only from {n, n-1, n-2} with arc weights <= 0, N <= 256, C <= 64, Bg == B — the lattices
 * CTCGraphGen builds for mono-character transcripts (fst_utils.py:603-613; the `2 L + 1`
 * CTC chain).  One wave per direction keeps four consecutive states per lane (no
 * transcendental, no LDS and no barrier on the recurrence), alpha / beta are block floating
 * point (one binary exponent per lane, rescaled by exact powers of two), the posterior rows
 * are normalised by Z (every group of four rows is checked to sum to four); only every second
 * alpha / beta row goes through the workspace (the consumer recomputes the one in between).
 *   grad_sign   +1, or -1 for the occupancies of -logZ (see asr_lattice_fwbw_signed_f32; ABI v20)
 *   ctc_labels  NULL, or the transcripts [B, ctc_lmax] (+ ctc_label_lens [B]) the graph matrices were
 *               built from by asr_ctc_graph_build with context_order 1 (N >= 2 ctc_lmax + 1): the
 *               kernel then writes the chain down from the labels and reads the matrices only for
 *               utterances the log-domain body has to redo (ABI v21)
 *   redo_count  device word of the caller or NULL: incremented once per utterance that the
 *               log-domain body had to redo; never reset by the library (a running counter:
 *               the caller takes differences).  (ABI v17)
 * asr_lattice_fwbw_band_supported says whether the SHAPES qualify; the graph of every
 * utterance is checked inside the kernel, and an utterance whose graph has another shape,
 * has no feasible alignment, or whose numbers leave the fp32 range runs the generic
 * log-domain body in the same launch (correct, slow): route here only graphs known to be
 * band-shaped (att_speech._native tags the ones its builders make and checks the others on
 * the host).  ASR_EUNSUPPORTED when the shapes do not qualify.
 */
int asr_lattice_fwbw_band_supported(int T, int B, int C, int N, int Kin, int Kout, int Bg);
int asr_lattice_fwbw_band_f32(const float *lp, int T, int B, int C,
                              const int32_t *lens,
                              const int32_t *src_in, const int32_t *il_in,
                              const float *w_in, const float *term,
                              const int32_t *dst_out, const int32_t *il_out,
                              const float *w_out,
                              int N, int Kin, int Kout, int Bg, float neg_inf, float grad_sign,
                              float *out_logZ, float *out_grad,
                              float *out_logZ_bwd,
                              void *workspace, int64_t workspace_bytes,
                              uint32_t *redo_count,
                              const int32_t *ctc_labels, const int32_t *ctc_label_lens, int ctc_lmax,
                              void *stream);

/*
 * Alpha-only scan: path_reduction's autodiff branch evaluated forward
 * (fst_utils.py:349-397) with reduction logsumexp (viterbi == 0) or max
 * (viterbi == 1, fst_utils.py:366-370).
 * For viterbi == 1 and out_best_il != NULL it also returns the input label of
 * the best path's arc at every frame — what FSTDecoder.decode reads from the
 * autograd gradient as `logits.grad.min(-1)[1]` (advanced_decoder.py:546-554);
 * rows t >= lens[b] are 0.  Ties pick the first maximum.
 *   out_score   [B]   f32
 *   out_best_il [T,B] i32 or NULL
 *   workspace: asr_lattice_viterbi_workspace_bytes(T,B,N) bytes when
 *              out_best_il != NULL, else may be NULL.
 */
int64_t asr_lattice_viterbi_workspace_bytes(int T, int B, int N);

int asr_lattice_forward_f32(const float *lp, int T, int B, int C,
                            const int32_t *lens,
                            const int32_t *src_in, const int32_t *il_in,
                            const float *w_in, const float *term,
                            int N, int K, int Bg, float neg_inf, int viterbi,
                            float *out_score, int32_t *out_best_il,
                            void *workspace, int64_t workspace_bytes,
                            void *stream);

/*
 * Row-wise log-softmax over contiguous groups: x viewed as [rows, group],
 * y = x - logsumexp(x, -1).  Replaces get_normalized_acts
 * (modules/ctc_losses.py:29-43): group = C for the plain branch (:41-42),
 * group = num_symbols for normalize_by_dim = context_order-1 (:34-40,
 * the per-context block-wise normalisation over the last symbol axis).
 */
int asr_log_softmax_fwd_f32(const float *x, int64_t rows, int group,
                            float *y, void *stream);

/* dx = dy - exp(y) * sum(dy, -1)   (autograd of the above) */
int asr_log_softmax_bwd_f32(const float *y, const float *dy, int64_t rows,
                            int group, float *dx, void *stream);

/*
 * FSTDecoder.get_fst_loss's stabilisation (advanced_decoder.py:479-484):
 *   row_max[t,b] = max_c x[t,b,c];  y = x - row_max;
 *   max_sum[b]   = sum_{t < lens[b]} row_max[t,b]
 * x, y [T,B,C]; row_max [T,B] (required, also the reduction scratch);
 * max_sum [B], summed in a fixed order (bitwise reproducible).
 */
int asr_sub_rowmax_f32(const float *x, int T, int B, int C,
                       const int32_t *lens, float *y, float *row_max,
                       float *max_sum, void *stream);

/*
 * Index of the first maximum of every row of x [rows, C]: the per-frame
 * arg-max of CTCDecoderAdvanced.decode (advanced_decoder.py:352,
 * `torch.max(logits_t, 2)`).  out_idx [rows] i32.
 */
int asr_argmax_rows_f32(const float *x, int64_t rows, int C, int32_t *out_idx,
                        void *stream);

/*
 * Device-side construction of the CTC training lattices for a batch of label
 * sequences (context orders 1 and 2), directly in the int32 / f32 layout the
 * lattice entry points take ([B, N = 2*Lmax+1, K = 3], padding arcs at
 * nc_weight).  Replaces get_training_matrices_batch (fst_utils.py:607-613:
 * OpenFst compose per utterance + fst_to_matrices :222-294 +
 * batch_training_graph_matrices :491-521) and the per-step host-to-device copy
 * of the 8 padded tensors (advanced_decoder.py:457-459).
 *   labels [B,Lmax] i32 symbols in [1, num_symbols), already reduced modulo
 *   num_symbols for bigram data sets (fst_utils.py:595-600); label_lens [B].
 */
int asr_ctc_graph_build(const int32_t *labels, const int32_t *label_lens, int B, int Lmax,
                        int num_symbols, int context_order, int allow_nonblank_selfloops,
                        int use_contextual_blanks, float nc_weight, int32_t *src_in,
                        int32_t *il_in, float *w_in, float *term, int32_t *dst_out,
                        int32_t *il_out, float *w_out, void *stream);

/*
 * Forward-backward / alpha scan over GROUP-FACTORED graphs: the reference's CTC
 * decoding graphs (build_ctc_mono_decoding_fst fst_utils.py:679-726,
 * build_ctc_bigram_decoding_fst :729-835), i.e. the denominator graph of the
 * globally normalised loss (advanced_decoder.py:473,497-500) and the search
 * graph of FSTDecoder.decode (:542-554).  State s1 feeds group g_of[s1]; state
 * s2 accepts an arc from every s1 with g_of[s1] == h_of[s2], plus an extra
 * self-loop when selfx[s2]; every arc into s2 consumes label[s2] with weight 0;
 * the start state is 0.  Same results as asr_lattice_fwbw_f32 /
 * asr_lattice_forward_f32 on the equivalent padded arc matrices (up to fp32
 * summation order) at ~1/17 of the arithmetic for the 2401-state bigram graph.
 *   g_of, h_of, label, selfx, uniq [N] i32 (uniq: no other state has this label)
 *   mem_g [G,Wg], mem_h [G,Wh] i32: states of every group, ascending, -1 padded
 *   term [N] f32 terminal log-weights
 * Outputs / workspace as for the generic entry points;
 * workspace: asr_lattice_grouped_workspace_bytes(T,B,N,G).  Needs 16*G <= 1024.
 */
int64_t asr_lattice_grouped_workspace_bytes(int T, int B, int N, int G);

int asr_lattice_grouped_fwbw_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, float *out_logZ, float *out_grad, float *out_logZ_bwd, void *workspace,
    int64_t workspace_bytes, void *stream);
/* ... with accumulate != 0: out_grad += occupancies (rows past an utterance's end are left as they
 * are) — the denominator of FSTDecoder.get_fst_loss (advanced_decoder.py:497-500) added onto the
 * numerator's (negated) occupancies in the same buffer: d loss / d acts without a separate
 * [T,B,C] addition (ABI v22). */
int asr_lattice_grouped_fwbw_acc_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, int accumulate, float *out_logZ, float *out_grad, float *out_logZ_bwd,
    void *workspace, int64_t workspace_bytes, void *stream);

int asr_lattice_grouped_forward_f32(
    const float *lp, int T, int B, int C, const int32_t *lens, int N, int G, int Wg, int Wh,
    const int32_t *g_of, const int32_t *h_of, const int32_t *label, const int32_t *selfx,
    const int32_t *uniq, const int32_t *mem_g, const int32_t *mem_h, const float *term,
    float neg_inf, int viterbi, float *out_score, int32_t *out_best_il, void *workspace,
    int64_t workspace_bytes, void *stream);

/*
 * Bidirectional, bias-free LSTM recurrence on a padded, length-masked batch —
 * the sequential part of BatchRNN (modules/encoders/encoder_utils.py:55-124:
 * nn.LSTM(bidirectional=True, bias=False) applied to a PackedSequence).
 * The caller supplies the input projection of every frame,
 *   gx [T,B,2,4H] f32, or bf16 with gx_bf16 != 0 (what a bf16 GEMM emits; halves its
 *      1.75 GB output at B=512) = x · W_ihᵀ   (direction-major, gate order i,f,g,o),
 * and the recurrent weights as bf16 (MFMA operands; accumulation, gate arithmetic and
 * the cell state are fp32; the gates SAVED for the backward pass are rounded to bf16):
 * whh [2,4H,H] for the forward pass,
 * whhT [2,H,4H] (transposed) for the backward pass.
 * Utterance b is active at frame t iff t < lens[b]; padding frames emit zeros
 * and carry no gradient, which reproduces pack_padded_sequence semantics
 * (the reverse direction starts at each utterance's own last frame).
 *   y      [T,B,2,H]   per-direction hidden outputs (the reference sums the
 *                      two directions, encoder_utils.py:112-117); may be null when the
 *                      caller only needs the bf16 copy below (inner layers of a stack)
 *   y_bf16 [2,T+2,B,H] bf16 copy, direction-major, frame t at index t+1 with a
 *                      zero frame at both ends: h_{t-1} of the forward
 *                      direction is the slice [0][0:T], of the reverse
 *                      direction [1][2:T+2] — contiguous GEMM operands for dW_hh
 *   gates_bf16 [T,2,B,H,4] bf16 post-activation gates (one 8-byte record i,f,g,o per
 *                      hidden unit), csave [T,2,B,H] f32 cell states: saved by the
 *                      forward pass for the backward pass
 *   dy     [T,B,2,H]   gradient w.r.t. y; dy_shared != 0: [T,B,H], one gradient for
 *                      both directions (BatchRNN sums them, encoder_utils.py:112-117)
 *   dgates_bf16 [T,B,2,4H] bf16: gradient w.r.t. the gate pre-activations
 *                      (= w.r.t. gx); the caller forms dx, dW_ih, dW_hh from it
 *                      with dense GEMMs (bf16 operands, fp32 accumulation)
 * workspace: asr_lstm_workspace_bytes(B, H) bytes.  Hidden sizes built:
 * 64, 128, 256, 320, 384, 512, 768 (ASR_EUNSUPPORTED otherwise).
 *   err_flag  optional device word owned by the caller (never cleared by the library).
 *             The persistent recurrence hands tiles between co-resident workgroups with
 *             bounded spins; if a spin bound expires (another stream's kernel kept a team
 *             mate off the device) the kernel stores 1 here and poisons its outputs with
 *             NaN instead of hanging.  A caller that reads a non-zero word must discard
 *             the step (att_speech._native.lstm_check_errors raises).  Debug knob:
 *             ASR_LSTM_SPIN_LIMIT=<n> overrides the bound (0 forces the timeout).
 */
int64_t asr_lstm_workspace_bytes(int B, int H);

int asr_lstm_bidir_fwd_bf16(const void *gx, int gx_bf16, const void *whh_bf16,
                            const int32_t *lens, int T, int B, int H,
                            float *y, void *y_bf16, void *gates_bf16, float *csave,
                            void *workspace, int64_t workspace_bytes,
                            uint32_t *err_flag, void *stream);

/* Weight gradients of one layer from the gate gradients, in one pass over them (replaces the
 * three chunked library products + sums of the reference's autograd for nn.LSTM weights,
 * encoder_utils.py:78,100):
 *   dgates_bf16 [T*B, 2, 4H]  as written by asr_lstm_bidir_bwd_bf16
 *   x_bf16      [T*B, H] layer input (input size == H), or NULL: then dw_ih must be NULL too
 *               and only dw_hh is computed (first layer: the caller multiplies for dw_ih)
 *   y_bf16      [2, T+2, B, H] the forward call's zero-padded bf16 outputs (h_{t-1} operands)
 *   dw_ih [2*4H, H], dw_hh [2][4H][H]  f32 out (overwritten)
 * Built for H = 320 (asr_lstm_wgrad_supported; ASR_EUNSUPPORTED otherwise).  fp32
 * accumulation over all frames in a fixed order: deterministic. */
int asr_lstm_wgrad_supported(int H);
int64_t asr_lstm_wgrad_workspace_bytes(int T, int B, int H, int with_input);
int asr_lstm_wgrad_bf16(const void *dgates_bf16, const void *x_bf16, const void *y_bf16,
                        int T, int B, int H, float *dw_ih, float *dw_hh,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* Input gradient of one layer from the gate gradients: dx [T*B, H] f32 = dgates [T*B, 2*4H] ·
 * w_ih [2*4H, H] (both bf16, w_ih = nn.LSTM weight_ih_l0 / _reverse stacked; input size == H).
 * Replaces the library product behind asr_lstm_bidir_bwd_bf16.  Built for H = 320. */
int asr_lstm_dgrad_supported(int H);
int asr_lstm_dgrad_bf16(const void *dgates_bf16, const void *w_ih_bf16, int T, int B, int H,
                        float *dx, void *stream);

/* The same layer with the input projection fused into the persistent recurrence
 * (replaces the `x·W_ihᵀ` GEMM in front of asr_lstm_bidir_fwd_bf16 when the layer's input
 * size F equals H, i.e. every BatchRNN after the first, encoder_utils.py:97-124 — or, for
 * H = 320, F = 352: the first layer behind the reference's conv front-end):
 *   x_bf16    [T,B,F] bf16 row-major layer input
 *   wih_bf16  [2 dir][4H][F] bf16, rows in gate order i,f,g,o (nn.LSTM weight_ih_l0 /
 *             weight_ih_l0_reverse stacked)
 *   y         must be NULL when F != H (that variant writes y_bf16 only)
 * Pre-activations are accumulated in fp32 over both products (no bf16 rounding of
 * x·W_ihᵀ).  ASR_EUNSUPPORTED when bit 0 of asr_lstm_fused_supported(B, H, F) is clear
 * (hidden size not one of 64/128/256/320, another input size, persistent path switched
 * off): the caller then uses the GEMM + asr_lstm_bidir_fwd_bf16.  Bits 1 and 2 of the same
 * word describe the two variants of include/asr_amd_experiments.h (not in the default build). */
int asr_lstm_fused_supported(int B, int H, int F);
int asr_lstm_bidir_fwd_fused_bf16(const void *x_bf16, const void *wih_bf16,
                                  const void *whh_bf16, const int32_t *lens,
                                  int T, int B, int H, int F, float *y, void *y_bf16,
                                  void *gates_bf16, float *csave, void *workspace,
                                  int64_t workspace_bytes, uint32_t *err_flag, void *stream);

int asr_lstm_bidir_bwd_bf16(const float *dy, int dy_shared, const void *whhT_bf16,
                            const int32_t *lens, int T, int B, int H,
                            const void *gates_bf16, const float *csave,
                            void *dgates_bf16,
                            void *workspace, int64_t workspace_bytes,
                            uint32_t *err_flag, void *stream);

/*
 * BatchNorm2d + Hardtanh(lo, hi) fused over the [B, C, H, W] output of a
 * convolution (fp32, or bf16 with x_bf16 — channels-last only; dx then is bf16 too): the `Normalization('batch_norm')` + `nn.Hardtanh(0, 20)` pair of
 * the DeepSpeech2 conv front-end (deep_speech_2.py:60-73).  training != 0: batch
 * statistics (biased variance for the normalisation; running_mean / running_var
 * updated with `momentum` and the unbiased variance like nn.BatchNorm2d, pass
 * null to leave them alone); training == 0: running statistics.  The clamped
 * activation is written as fp32 or bf16 (out_bf16), in NCHW or time-major
 * [H, B, C, W] order (out_time_major: the permute(2,0,1,3) of
 * deep_speech_2.py:142-146).  channels_last != 0: x (and a non-time-major out /
 * dy, and dx) are stored [B, H, W, C] — the layout MIOpen's implicit-GEMM
 * convolutions produce and consume (C % 4 == 0 and C/4 must divide 256).  save_mean / save_invstd
 * [C] feed the backward.  conv_bias [C] (or null): the bias of the convolution that
 * produced x, added on the fly (x is then the bias-free convolution), so the
 * framework needs neither the broadcast add nor the full-tensor reduction for its
 * gradient; the backward returns that gradient in dconv_bias (may be null).
 * chan_sums (or null): [2][C] doubles, (sum x, sum x^2) per channel of the bias-free input,
 * as the convolution kernels below emit them from their epilogues — the statistics pass
 * over x is then skipped (training only).
 * workspace: asr_bn_act_workspace_bytes(C).
 */
int64_t asr_bn_act_workspace_bytes(int C);
int asr_bn_act_fwd_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                       const float *gamma, const float *beta,
                       float *running_mean, float *running_var,
                       int channels_last,
                       int training, float momentum, float eps, float lo, float hi,
                       void *out, int out_bf16, int out_time_major,
                       float *save_mean, float *save_invstd,
                       const double *chan_sums,
                       void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Backward of the above from the SAVED convolution output x: the Hardtanh mask
 * (lo < bn(x) < hi, torch's hardtanh_backward) is recomputed, dy is read in the
 * dtype / layout the forward wrote.  dx [B,C,H,W] fp32, dgamma / dbeta [C].
 */
int asr_bn_act_bwd_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                       const float *gamma, const float *beta,
                       const float *save_mean, const float *save_invstd,
                       int channels_last,
                       int training, float lo, float hi,
                       const void *dy, int dy_bf16, int dy_time_major,
                       void *dx, float *dgamma, float *dbeta, float *dconv_bias,
                       void *workspace, int64_t workspace_bytes, void *stream);

/*
 * The same backward in two halves, for replicas that share their batch statistics (SyncBN-style
 * data parallelism, att_speech.dp.enable_sync_batchnorm; the single-process reference normalises
 * over the whole batch, deep_speech_2.py:21,60-73) (ABI v19).  phase 1: the per-channel sums
 * {sum dy', sum dy' xhat} of THIS replica as 2 C interleaved doubles at the start of the workspace,
 * and dgamma / dbeta / dconv_bias from them (parameter gradients are local: the gradient
 * all-reduce adds the replicas up); phase 2: dx from the sums the caller left there — all-reduced
 * over the replicas — and n_total, the number of elements per channel they stand for.  phase 0 =
 * asr_bn_act_bwd_f32.
 */
int asr_bn_act_bwd_phase_f32(const void *x, int x_bf16, const float *conv_bias, int B, int C, int H, int W,
                             const float *gamma, const float *beta,
                             const float *save_mean, const float *save_invstd,
                             int channels_last,
                             int training, float lo, float hi,
                             const void *dy, int dy_bf16, int dy_time_major,
                             void *dx, float *dgamma, float *dbeta, float *dconv_bias,
                             void *workspace, int64_t workspace_bytes, int phase, double n_total,
                             void *stream);

/*
 * log_softmax over all C classes followed by the per-frame max stabilisation of
 * FSTDecoder.get_fst_loss (advanced_decoder.py:444-452 with normalize_by_dim = 0, then
 * :479-484) in ONE pass over x [T, B, C]: y = log_softmax(x) - max_c log_softmax(x) = x - max_c x,
 * nls [T, B] = max_c log_softmax(x) (= -log sum_c exp(x - max x)), nls_sum [B] = sum over the
 * frames t < lens[b] of nls — the value the reference subtracts as its denominator when
 * denominator_red = 'none'.  bwd: dx = dy - exp(y + nls) * sum_c dy, the gradient of the
 * log-softmax (the reference detaches the maximum), from the shifted y itself.
 */
int asr_log_softmax_shift_fwd_f32(const float *x, int T, int B, int C, const int32_t *lens,
                                  float *y, float *nls, float *nls_sum, void *stream);
int asr_log_softmax_shift_bwd_f32(const float *y, const float *nls, const float *dy, int64_t rows,
                                  int C, float *dx, void *stream);

/* asr_log_softmax_shift_bwd_f32 for the split-bf16 class projection behind it (wide alphabets):
 * the gradient leaves as dx = hi + lo, two bf16 tensors [rows, ld] (ld >= C, ld <= 2560; columns
 * >= C zero) — the fp32 tensor is never written — together with its column sums (the
 * projection's bias gradient) as asr_log_softmax_shift_bwd_split_blocks(rows) partial rows
 * colsum_partial [blocks, ld], which the caller adds up (asr_sum_leading_f32).  (ABI v15) */
int asr_log_softmax_shift_bwd_split_blocks(int64_t rows);
int asr_log_softmax_shift_bwd_split_bf16(const float *y, const float *nls, const float *dy,
                                         int64_t rows, int C, void *hi_bf16, void *lo_bf16, int ld,
                                         float *colsum_partial, void *stream);

/*
 * out[e] = sum over g < G of in[g * n + e] (n % 4 == 0): the sum of the partial products of a
 * weight-gradient GEMM split over chunks of frames (torch's strided reduction reads at
 * 1.4 TB/s here).
 */
int asr_sum_leading_f32(const float *in, int G, int64_t n, float *out, void *stream);

/*
 * x = hi + lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi): the operands of the
 * split-bf16 class projection of the decoders (advanced_decoder.py:79-223 computes
 * `F.linear(frames, weight, bias)` in fp32; with C = 2401 classes the three bf16 MFMA products
 * hi hi + hi lo + lo hi, accumulated in fp32, stand for it at 2^-16 relative error per term).
 *   x [rows, cols] f32 with row stride ldx (elements); hi / lo [rows, cols] bf16 with row
 *   strides ldhi / ldlo — strided so that the halves can be written straight into the
 *   K-concatenated operand [rows, 3 cols].  A contiguous tensor is one row of n elements.
 * (ABI v14)
 */
int asr_split_bf16_f32(const float *x, int64_t rows, int64_t cols, int64_t ldx,
                       void *hi_bf16, int64_t ldhi, void *lo_bf16, int64_t ldlo, void *stream);

/*
 * The 7x7, 32 -> 32 channel convolution of the DeepSpeech2 front-end (reference
 * att_speech/modules/encoders/deep_speech_2.py:60-73, Conv2d(32, 32, (7, 7), stride
 * (stride_h, 1)), stride_h in {1, 3}) on channels-last bf16 with fp32 accumulation, bias-free
 * (the bias is folded into asr_bn_act_*):
 *   x [B, H, W, 32] bf16, w [32, 32, 7, 7] f32 (nn.Conv2d layout), y [B, Ho, Wo, 32] bf16,
 *   Ho = (H - 7) / stride_h + 1, Wo = W - 6 (<= 48).
 * fwd: y = conv(x, w);  bwd_data: dx = conv_transpose(dy, w) [B, H, W, 32] bf16;
 * wgrad: dw [32, 32, 7, 7] f32 = sum over the batch of dy (x) x (overwritten).
 * chan_sums (or null): [2][32] doubles, per output channel the sum and the sum of squares of
 * the bf16 outputs — what the following BatchNorm needs (asr_bn_act_fwd_f32).
 * workspace: asr_conv7x7c32_workspace_bytes() (packed weight fragments / partial sums).
 */
int64_t asr_conv7x7c32_workspace_bytes(void);
int asr_conv7x7c32_fwd_bf16(const void *x, const float *w, int B, int H, int W, int stride_h,
                            void *y, double *chan_sums, void *workspace, int64_t workspace_bytes,
                            void *stream);
/* stride_h = 3 only (the shape the encoder uses); H, W are the INPUT's (dx's) extents */
int asr_conv7x7c32_bwd_data_bf16(const void *dy, const float *w, int B, int H, int W,
                                 int stride_h, void *dx, void *workspace,
                                 int64_t workspace_bytes, void *stream);
int asr_conv7x7c32_wgrad_bf16(const void *x, const void *dy, int B, int H, int W, int stride_h,
                              float *dw, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * The first convolution of the front-end, Conv2d(1, 32, (7, 7), stride (1, 2), padding (6, 0))
 * (reference deep_speech_2.py:52-66) on the raw one-channel features:
 *   x [B, T, F] f32, w [32, 1, 7, 7] f32, y [B, T + 6, (F - 7) / 2 + 1, 32] bf16 channels-last,
 *   bias-free; wgrad: dw [32, 1, 7, 7] f32 from x and dy (y's layout).  The input needs no
 *   gradient.  workspace: asr_conv1_7x7s2_workspace_bytes().
 */
int64_t asr_conv1_7x7s2_workspace_bytes(void);
int asr_conv1_7x7s2_fwd(const float *x, const float *w, int B, int T, int F, void *y,
                        double *chan_sums, void *workspace, int64_t workspace_bytes, void *stream);
int asr_conv1_7x7s2_wgrad(const float *x, const void *dy, int B, int T, int F, float *dw,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* The same first convolution for the WSJ recipes' real features: x [B, T, F, cin] f32 — the
 * reference's bs x t x f x c layout (deep_speech_2.py:127), cin = 3 (static, delta,
 * delta-delta; egs/wsj/yamls/ctc.yaml:8-15), F = 81 -> Fo = 38 — w [32, cin, 7, 7] f32, the
 * same outputs.  Built for cin == 3 and even Fo <= 48 (ASR_EUNSUPPORTED otherwise); workspace
 * asr_conv1c_7x7s2_workspace_bytes(cin).  (ABI v16) */
int64_t asr_conv1c_7x7s2_workspace_bytes(int cin);
int asr_conv1c_7x7s2_fwd(const float *x, const float *w, int B, int T, int F, int cin, void *y,
                         double *chan_sums, void *workspace, int64_t workspace_bytes, void *stream);
int asr_conv1c_7x7s2_wgrad(const float *x, const void *dy, int B, int T, int F, int cin,
                           float *dw, void *workspace, int64_t workspace_bytes, void *stream);

/*
 * One label step of the TCN / local-attention decoder for every live hypothesis
 * (replaces LocalAttention.forward, reference att_speech/modules/tcn.py:193-230, and the
 * context reduction of AttentionDecoderTCN.enc_step, :465-474).  Hypothesis h = u * beam + k
 * belongs to utterance u.
 *   eproj   [T, B, A]   encoded_to_hidden(encoded), per utterance, time-major
 *   enc     [T, B, E]   encoder output;  enc_lens [B]
 *   filt    [B*beam, A, Kf]  lm_to_kernel(lm_state) (Kf = 32 taps), glob [B*beam, A] lm_to_global(lm_state)
 *   w_score [A], b_score: hidden_to_score;  temperature: LocalAttention.temperature
 *   att_prev [B*beam, T] previous alignments, one row per hypothesis; parent [B*beam] (or
 *           null): hypothesis h continues row parent[h] (the beam's re-indexing, :551)
 *   att_new [B*beam, T]  softmax_t( w . tanh(eproj_t + (a_prev * filt)(t) + glob) * temperature + pad_t )
 *   context [B*beam, E]  sum_t att_new[t] enc[t]
 */
int asr_tcn_attention_step_f32(const float *eproj, const float *enc, const int32_t *enc_lens,
                               const float *filt, const float *glob, const float *w_score,
                               float b_score, float temperature,
                               const float *att_prev, const int32_t *parent,
                               int T, int B, int beam, int A, int Kf, int E,
                               float *att_new, float *context, void *stream);

/*
 * One step of BeamSearch for every utterance, without a host read-back (replaces
 * BeamSearch.step + _save_best_finished, reference att_speech/modules/beam_search.py:58-124,
 * 147-175).  logits [B*beam, C] (class C-1 = EOS); scores_in / scores_out [B*beam] running
 * scores (distinct buffers); est_in / est_out [B*beam, Lcap] label histories (distinct
 * buffers, `step` labels each on entry); len_div = step ** length_normalization.
 * Per utterance: finished_count [B], best_score [B] (init -inf; the RAW EOS score, the
 * reference's aliasing quirk), best_len [B], best_tokens [B, Lcap].  Outputs new_input
 * [B*beam] (chosen labels), parent [B*beam] (flat index of the hypothesis each survivor
 * extends).  done_and_scratch [3] int32, zero before the first step: word 0 becomes 1 after
 * the step in which every utterance reached finished_count >= beam; calls with the flag set
 * change nothing, so a host polling it every few steps sees the reference's results; word 2
 * counts the steps that took effect (the final histories are in the buffer written by the
 * last of them), word 1 is scratch.
 * Ties in the top-k go to the lower candidate index.  beam <= 32, beam * (C-1) <= 2048.
 */
int asr_beam_step_f32(const float *logits, const float *scores_in, float *scores_out,
                      const int32_t *est_in, int32_t *est_out, int step, int B, int beam, int C,
                      int Lcap, float len_div, int32_t *finished_count, float *best_score,
                      int32_t *best_len, int32_t *best_tokens, int32_t *new_input,
                      int32_t *parent, int32_t *done_and_scratch, void *stream);

/*
 * The step boundary on the device (ABI v18): the reference's GradientClipping hook
 * (att_speech/modules/hooks/gradient_clipping.py:13-53: clip_grad_norm_ to clip_norm, skip the
 * optimizer step when the unclipped norm exceeds skip_step_norm) and torch.optim.Adam.step
 * (trainer.py:262-266; amsgrad / maximize off, weight_decay as L2) without a host read-back
 * between backward and the update.
 *
 * asr_grad_sumsq_partials_f32: partials[i] = sum of g^2 over the i-th of nparts equal slices of
 * the flat gradient g [n] (nparts <= 65535).
 * asr_adam_clip_step_f32: norm = sqrt(sum partials); the step is skipped when norm >
 * skip_norm, when it is not finite, or when *err_word != 0 (the persistent LSTM's error word,
 * may be NULL); otherwise g is scaled by min(1, clip_norm / (norm + 1e-6)) and Adam step number
 * *step_in + 1 updates m_flat / v_flat [n] and the parameters.  The parameters stay where the
 * caller's framework allocated them: `chunks` (device memory, nchunks entries)
 * lists pieces of at most asr_adam_chunk_elems() elements: `param` = address of the piece
 * inside its parameter tensor, `flat_offset` = its offset in g / m / v.  *step_out = *step_in + 1
 * (or *step_in when skipped); step_in != step_out (the caller alternates two words).
 * stats [4] = {norm, clipped, skipped, err} as floats, for the host to read when it likes.
 */
/*
 * asr_lattice_fwbw_f32 with the occupancies written times grad_sign (+1 or -1) (ABI v20): a
 * decoder that needs -logZ (the numerator of FSTDecoder.get_fst_loss, advanced_decoder.py:486-496)
 * asks for -1 and receives the gradient of what it uses; the backward pass of PathLogSumExp
 * (fst_utils.py:482-485: grad_output[None, :, None] * grads) then multiplies by +1 —
 * asr_scale_rows_f32 (x [T,B,C] *= scale [B] in place) leaves utterances whose factor is exactly 1
 * untouched, so the usual backward pass makes no pass over the [T,B,C] tensor at all (1.6 GB at
 * C = 2401, 512 utterances).  out_logZ is +logZ either way.
 */
int asr_lattice_fwbw_signed_f32(const float *lp, int T, int B, int C, const int32_t *lens,
                                const int32_t *src_in, const int32_t *il_in, const float *w_in,
                                const float *term, const int32_t *dst_out, const int32_t *il_out,
                                const float *w_out, int N, int Kin, int Kout, int Bg, float neg_inf,
                                float grad_sign, float *out_logZ, float *out_grad,
                                float *out_logZ_bwd, void *workspace, int64_t workspace_bytes,
                                void *stream);
int asr_scale_rows_f32(float *x, int T, int B, int C, const float *scale, void *stream);

typedef struct AsrAdamChunk {
    void *param;
    uint32_t flat_offset;
    uint32_t count;
} AsrAdamChunk;
int asr_adam_chunk_elems(void);
int asr_grad_sumsq_partials_f32(const float *g, int64_t n, float *partials, int nparts, void *stream);
int asr_adam_clip_step_f32(const AsrAdamChunk *chunks, int nchunks, const float *g_flat,
                           float *m_flat, float *v_flat, const float *partials, int nparts,
                           const uint32_t *err_word, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float clip_norm, float skip_norm,
                           const int32_t *step_in, int32_t *step_out, float *stats, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_AMD_H */
