/*
 * include/asr_amd_experiments.h — entry points that are NOT part of libasr_amd.so's ABI.
 *
 * Two variants of the BiLSTM recurrence that were measured and lost against the default path
 * (the fused-dx backward is slower than recurrence + dgrad GEMM; the in-kernel
 * direction sum is break-even).  They are kept as compiled-on-demand experiments: the default
 * build does not export them, tests/test_cabi.py checks that it does not, and nothing in the
 * product path calls them unless the library was built with
 *
 *     make -C pytorch-asr_amd/csrc EXPERIMENTS=1        (adds -DASR_EXPERIMENTS)
 *
 * and the matching opt-in switch is set (ASR_LSTM_DIRSUM=1 / ASR_LSTM_FUSED_BWD=1).
 * Conventions as in asr_amd.h.
 */
#ifndef ASR_AMD_EXPERIMENTS_H
#define ASR_AMD_EXPERIMENTS_H

#include "asr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* asr_lstm_bidir_fwd_fused_bf16, and the direction merge of BatchRNN (encoder_utils.py:112-117: the sum of the two
 * directions is the next layer's input) on the bf16 planes:
 *   xsum_bf16 [T,B,H] bf16 = bf16(h_fwd) + bf16(h_rev) (the value of y_bf16[0,t+1] + y_bf16[1,t+1]
 *             rounded to bf16, bit for bit).
 * The recurrence runs as two launches, steps [0, ceil(T/2)) and the rest: in the second one
 * every frame a direction reaches was written by the other direction in the first, so it adds
 * that value to its own output on the way out — no extra pass over the planes.  For odd T the
 * middle frame xsum[T/2] is NOT written (both directions reach it in the first launch): the
 * caller adds that one frame.  ASR_EUNSUPPORTED as above, for T < 2 and when bit 2 of
 * asr_lstm_fused_supported(B, H, F) is clear (H != 320; the 352-feature layer at 32-row batch
 * tiles). */
int asr_lstm_bidir_fwd_fused_sum_bf16(const void *x_bf16, const void *wih_bf16,
                                      const void *whh_bf16, const int32_t *lens,
                                      int T, int B, int H, int F, float *y, void *y_bf16,
                                      void *gates_bf16, float *csave, void *xsum_bf16,
                                      void *workspace, int64_t workspace_bytes,
                                      uint32_t *err_flag, void *stream);

/* Backward recurrence with the input gradient fused (replaces the `dgates·W_ih` GEMM behind
 * asr_lstm_bidir_bwd_bf16 when the layer's input size equals H):
 *   wihT_bf16 [2 dir][H (input feature)][4H] bf16 — W_ih of each direction, transposed
 *   dx        [2 dir][T,B,H] f32 out: dgates_dir · W_ih_dir; the layer's input gradient is
 *             the sum of the two planes.  The layer below takes them as they are:
 *   dy_shared = 2 (both backward entry points): dy is such a pair of planes [2][T,B,H] and
 *             the shared output gradient is their sum (persistent kernels only).
 * asr_lstm_fused_supported returns bit 0 for the forward, bit 1 for this entry point (the
 * batch must fit one launch of 16- or 24-row tiles); ASR_EUNSUPPORTED otherwise. */
int asr_lstm_bidir_bwd_fused_bf16(const float *dy, int dy_shared, const void *whhT_bf16,
                                  const void *wihT_bf16, const int32_t *lens,
                                  int T, int B, int H, const void *gates_bf16,
                                  const float *csave, void *dgates_bf16, float *dx,
                                  void *workspace, int64_t workspace_bytes,
                                  uint32_t *err_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif
