"""Bidirectional bias-free LSTM layer on the MI355X: dense projections as bf16
GEMMs with fp32 accumulation, the recurrence in the hand-written per-step MFMA
kernels of csrc/lstm.hip (include/asr_amd.h: asr_lstm_bidir_{fwd,bwd}_bf16).

Replaces the vendor LSTM behind `BatchRNN.rnn` (reference
modules/encoders/encoder_utils.py:78,100) while keeping the nn.LSTM parameter
tensors (weight_ih_l0, weight_hh_l0, *_reverse), so state_dict keys and
optimizers are unchanged."""
import torch

from att_speech import _native


def _mm_f32(a, b):
    """bf16 x bf16 -> fp32 matmul (fp32 accumulate); falls back to rounding the
    product to bf16 where the runtime has no out_dtype."""
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.mm(a, b).float()


class BiLSTMFunction(torch.autograd.Function):
    """y[T,B,2,H] = BiLSTM(x[T,B,F]; W_ih[2][4H,F], W_hh[2][4H,H]), masked by lens."""

    @staticmethod
    def forward(ctx, x, lens_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r):
        T, B, F = x.shape
        H = w_hh_f.shape[1]
        xb = x.reshape(T * B, F).to(torch.bfloat16)
        w_ih = torch.cat([w_ih_f, w_ih_r], 0).to(torch.bfloat16)        # [2*4H, F]
        gx = _mm_f32(xb, w_ih.t()).view(T, B, 2, 4 * H)
        whh = torch.stack([w_hh_f, w_hh_r], 0).to(torch.bfloat16).contiguous()
        y, gates, csave = _native.lstm_bidir_fwd(gx, whh, lens_dev)
        ctx.save_for_backward(xb, lens_dev, w_ih, whh, y, gates, csave)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, lens_dev, w_ih, whh, y, gates, csave = ctx.saved_tensors
        T, B, _, H = y.shape
        F = xb.shape[1]
        whhT = whh.transpose(1, 2).contiguous()                          # [2,H,4H]
        dgates = _native.lstm_bidir_bwd(dy.contiguous(), whhT, lens_dev, gates, csave)
        dgb = dgates.view(T * B, 2 * 4 * H).to(torch.bfloat16)
        dx = _mm_f32(dgb, w_ih).view(T, B, F)
        dw_ih = _mm_f32(dgb.t(), xb)                                     # [2*4H, F]
        # h_{t-1}: forward direction looks one frame back, reverse one frame ahead
        hprev = torch.zeros_like(y)
        hprev[1:, :, 0] = y[:-1, :, 0]
        hprev[:-1, :, 1] = y[1:, :, 1]
        hb = hprev.to(torch.bfloat16)
        dg = dgb.view(T * B, 2, 4 * H)
        dw_hh_f = _mm_f32(dg[:, 0].t(), hb[:, :, 0].reshape(T * B, H))
        dw_hh_r = _mm_f32(dg[:, 1].t(), hb[:, :, 1].reshape(T * B, H))
        return dx, None, dw_ih[:4 * H], dw_hh_f, dw_ih[4 * H:], dw_hh_r


def bilstm(x, lens, rnn):
    """x [T,B,F] GPU tensor, lens [B] (any int tensor), rnn: nn.LSTM(bidirectional,
    bias=False, 1 layer).  Returns per-direction outputs [T,B,2,H]."""
    lens_dev = torch.as_tensor(lens).to(x.device, torch.int32)
    return BiLSTMFunction.apply(
        x.contiguous(), lens_dev, rnn.weight_ih_l0, rnn.weight_hh_l0,
        rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse)
