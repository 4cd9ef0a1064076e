"""Bidirectional bias-free LSTM layer on the MI355X: dense projections as bf16
GEMMs with fp32 accumulation, the recurrence in the hand-written persistent MFMA
kernels of csrc/lstm.hip (include/asr_amd.h: asr_lstm_bidir_{fwd,bwd}_bf16).

Replaces the vendor LSTM behind `BatchRNN.rnn` (reference
modules/encoders/encoder_utils.py:78,100) while keeping the nn.LSTM parameter
tensors (weight_ih_l0, weight_hh_l0, *_reverse), so state_dict keys and
optimizers are unchanged."""
import os

import torch

from att_speech import _native


def _mm_f32(a, b):
    """bf16 x bf16 -> fp32 matmul (fp32 accumulate); falls back to rounding the
    product to bf16 where the runtime has no out_dtype."""
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.mm(a, b).float()


# byte offsets inside the persistent kernels are 32-bit (csrc/lstm.hip, lstm_fwd_impl): gates
# [T,B,2,4H] f32-sized, the bf16 planes [2,T+2,B,H], the bf16 input [T,B,F]
_FUSED_LIMITS = {'gates': 2 ** 32, 'planes': 2 ** 32, 'input': 2 ** 31}


def _fused_fits(T, B, H, F):
    """asr_lstm_fused_supported knows the tile shapes but not T: a long batch (B=768 with
    T' >= 547 at H=320) exceeds the 32-bit offsets of the fused kernel, which then answers
    ASR_EUNSUPPORTED; such batches take the GEMM + recurrence path instead."""
    return (T * B * 8 * H * 4 < _FUSED_LIMITS['gates'] and 2 * (T + 2) * B * H * 2 < _FUSED_LIMITS['planes']
            and T * B * F * 2 < _FUSED_LIMITS['input'])


def _forward_layer(xb, w_ih, whh, lens_dev, T, B, H, want_y, want_sum=False):
    """One layer's recurrence from its bf16 input [T*B, F].  Where the library has the fused
    kernel (F == H; or H = 320, F = 352 when the fp32 outputs are not wanted) the input
    projection runs inside the persistent recurrence (asr_lstm_bidir_fwd_fused_bf16: no
    [T*B, 8H] product in HBM; ASR_LSTM_FUSED=0 restores the GEMM everywhere, =inner for the
    F != H layer only); otherwise x·W_ih is one GEMM, accumulated in fp32 and stored once as
    bf16 (ASR_GX_FP32=1 keeps fp32: the product is 1.75 GB in fp32 at B=512 and its write
    bounds the GEMM)."""
    F = xb.shape[1]
    mode = os.environ.get('ASR_LSTM_FUSED', '1')         # 0: never, inner: only F == H layers
    if (mode != '0' and (F == H or (not want_y and mode != 'inner'))
            and _native.lstm_fused_supported(B, H, F=F) and _fused_fits(T, B, H, F)):
        # want_sum + ASR_LSTM_DIRSUM=1: the direction sum on the bf16 planes comes out of the
        # recurrence itself (a fifth return value).  Opt-in: measured break-even at B=768 — the
        # second launch's set-up and the per-step tile fetch cost what the saved pass over the
        # planes (0.1 ms per layer) gains.
        if (want_sum and T >= 2 and os.environ.get('ASR_LSTM_DIRSUM', '0') == '1'
                and _native.lstm_fused_supported(B, H, F=F, dirsum=True)):
            return _native.lstm_bidir_fwd_fused(xb.view(T, B, F), w_ih, whh, lens_dev, want_y=want_y,
                                                want_sum=True)
        return _native.lstm_bidir_fwd_fused(xb.view(T, B, F), w_ih, whh, lens_dev, want_y=want_y)
    if os.environ.get('ASR_GX_FP32', '0') == '1':
        gx = _mm_f32(xb, w_ih.t()).view(T, B, 2, 4 * H)
    else:
        gx = torch.mm(xb, w_ih.t()).view(T, B, 2, 4 * H)
    return _native.lstm_bidir_fwd(gx, whh, lens_dev, want_y=want_y)


def _input_gradient(dgb, w_ih, T, B, H, F, bf16_out=False):
    """dx [T,B,F] = dgates · W_ih: the hand-written product (csrc/lstm_dgrad.hip) for F == H
    where it is built (ASR_LSTM_DGRAD=0: library GEMM), the library GEMM otherwise; bf16_out:
    the gradient of a bf16 layer input (the first layer behind the conv front-end)."""
    if (not bf16_out and F == H and os.environ.get('ASR_LSTM_DGRAD', '1') != '0'
            and _native.lstm_dgrad_supported(H) and T * B * 16 * H < 2 ** 31 - 2 ** 20):
        return _native.lstm_dgrad(dgb.view(T, B, 2, 4 * H), w_ih)
    # K-major second operand: the library's kernel for it is faster here (probe 0.38 vs
    # 0.44 ms) than the one it picks for the row-major [8H, F] weight; the transpose is
    # a 1.6 MB copy
    wk = w_ih.t().contiguous().t()
    dg2 = dgb.view(T * B, 8 * H)
    return (torch.mm(dg2, wk) if bf16_out else _mm_f32(dg2, wk)).view(T, B, F)


def _chunks(n, target):
    """largest power of two <= target that divides n and leaves >= 2048 rows per chunk"""
    g = target
    while g > 1 and (n % g or n // g < 2048):
        g //= 2
    return g


def _bmm_f32(a, b):
    try:
        return torch.bmm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.bmm(a, b).float()


class BiLSTMFunction(torch.autograd.Function):
    """y[T,B,2,H] = BiLSTM(x[T,B,F]; W_ih[2][4H,F], W_hh[2][4H,H]), masked by lens;
    sum_dirs: return y.sum(2) [T,B,H] (BatchRNN's merge, encoder_utils.py:112-117) so
    the backward kernel reads the one shared gradient instead of an expanded copy."""

    @staticmethod
    def forward(ctx, x, lens_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r, sum_dirs=False):
        T, B, F = x.shape
        H = w_hh_f.shape[1]
        xb = x.reshape(T * B, F).to(torch.bfloat16)
        ctx.x_bf16 = x.dtype == torch.bfloat16        # then the input gradient is bf16 too
        w_ih = torch.cat([w_ih_f, w_ih_r], 0).to(torch.bfloat16)        # [2*4H, F]
        whh = torch.stack([w_hh_f, w_hh_r], 0).to(torch.bfloat16).contiguous()
        y, ybf, gates, csave = _forward_layer(xb, w_ih, whh, lens_dev, T, B, H, True)
        ctx.save_for_backward(xb, lens_dev, w_ih, whh, ybf, gates, csave)
        return y.sum(2) if sum_dirs else y

    @staticmethod
    def backward(ctx, dy):
        xb, lens_dev, w_ih, whh, ybf, gates, csave = ctx.saved_tensors
        _, T2, B, H = ybf.shape
        T = T2 - 2
        F = xb.shape[1]
        whhT = whh.transpose(1, 2).contiguous()                          # [2,H,4H]
        dgb = _native.lstm_bidir_bwd(dy.contiguous(), whhT, lens_dev, gates, csave)
        dx = _input_gradient(dgb, w_ih, T, B, H, F, bf16_out=ctx.x_bf16)
        # Weight gradients: [4H.. x TB] x [TB x F|H] with TB = T*B frames and a small
        # output.  As one GEMM the library fills 100-170 of 256 CUs (0.79 / 1.06 ms
        # at TB = 171k); split over G chunks of frames as a batched GEMM plus a sum
        # of the G partial products they take 0.38 / 0.43 ms; the one-pass kernel of
        # csrc/lstm_wgrad.hip replaces them where it is built (_weight_gradients).
        # h_{t-1}: the forward direction looks one frame back (frames 0..T-1 of its
        # zero-padded bf16 plane), the reverse one frame ahead (frames 2..T+1).
        dw_ih, dw_hh = _weight_gradients(dgb, xb, ybf, T, B, H, F)
        return dx, None, dw_ih[:4 * H], dw_hh[0], dw_ih[4 * H:], dw_hh[1], None


def _weight_gradients(dgb, xb, ybf, T, B, H, F):
    """dW_ih [2*4H, F] and dW_hh[d] [4H, H] from the gate gradients: the one-pass kernel
    asr_lstm_wgrad_bf16 where it is built (H = 320; ASR_LSTM_WGRAD=0 switches it off), the
    chunked library products otherwise — and for dW_ih of a layer whose input size is not H."""
    if os.environ.get('ASR_LSTM_WGRAD', '1') != '0' and _native.lstm_wgrad_supported(H) \
            and (T * B + 1024) * 16 * H < 2 ** 31:        # the kernel's 32-bit buffer offsets
        if F == H:
            dw_ih, dw_hh = _native.lstm_wgrad(dgb.view(T, B, 2, 4 * H), xb, ybf)
            return dw_ih, [dw_hh[0], dw_hh[1]]
        _, dw_hh = _native.lstm_wgrad(dgb.view(T, B, 2, 4 * H), None, ybf)
        TB = T * B
        g1 = _chunks(TB, 64)         # (64, not 16: see _weight_gradients_library)
        dw_ih = _native.sum_leading(_bmm_f32(dgb.view(g1, TB // g1, 8 * H).transpose(1, 2),
                                             xb.view(g1, TB // g1, F)))
        return dw_ih, [dw_hh[0], dw_hh[1]]
    return _weight_gradients_library(dgb, xb, ybf, T, B, H, F)


def _weight_gradients_library(dgb, xb, ybf, T, B, H, F):
    """the same as chunked library GEMMs (see BiLSTMFunction.backward)"""
    TB = T * B
    dg2 = dgb.view(TB, 8 * H)
    # (64 chunks for dw_ih: 16 filled 100-170 CUs with the 352-column output of the first layer
    #  and cost 2x at B=768 — 912 vs 569 us; no difference at B=576)
    g1, g2 = _chunks(TB, 64), _chunks(TB, 32)
    dw_ih = _native.sum_leading(_bmm_f32(dg2.view(g1, TB // g1, 8 * H).transpose(1, 2),
                                         xb.view(g1, TB // g1, F)))
    dgd = dgb.view(g2, TB // g2, 2, 4 * H)
    dw_hh = [_native.sum_leading(_bmm_f32(
        dgd[:, :, d].transpose(1, 2),
        (ybf[0, 0:T] if d == 0 else ybf[1, 2:T + 2]).reshape(g2, TB // g2, H)))
        for d in range(2)]
    return dw_ih, dw_hh


class BiLSTMStackFunction(torch.autograd.Function):
    """A stack of bidirectional LSTM layers whose directions are summed between layers
    (BatchRNN x N without normalisation / projection / residual, encoder_utils.py:97-124) as
    ONE autograd node: between layers only the bf16 hidden planes the recurrence writes
    anyway are read — layer l+1's GEMM operand is bf16(h_fwd) + bf16(h_rev) — so the inner
    layers write no fp32 outputs, and no fp32 direction sum or cast pass runs between them
    (0.17 ms per layer at B=576); the gradient travels between layers in fp32."""

    @staticmethod
    def forward(ctx, x, lens_dev, *weights):
        T, B, F = x.shape
        n = len(weights) // 4
        saved, xb = [], x.reshape(T * B, F).to(torch.bfloat16)
        ctx.x_bf16 = x.dtype == torch.bfloat16        # then the input gradient is bf16 too
        # bf16 operands of all layers, the two directions stacked: ONE multi-tensor copy into
        # views of the stacked buffers (cat + cast per matrix were 16 small launches per step)
        w_ihs, whhs, dst, src = [], [], [], []
        for l in range(n):
            w_ih_f, w_hh_f, w_ih_r, w_hh_r = weights[4 * l:4 * l + 4]
            H4, Fl = w_ih_f.shape
            w_ih = torch.empty((2 * H4, Fl), dtype=torch.bfloat16, device=x.device)
            whh = torch.empty((2, H4, w_hh_f.shape[1]), dtype=torch.bfloat16, device=x.device)
            dst += [w_ih[:H4], w_ih[H4:], whh[0], whh[1]]
            src += [w_ih_f.detach(), w_ih_r.detach(), w_hh_f.detach(), w_hh_r.detach()]
            w_ihs.append(w_ih)
            whhs.append(whh)
        torch._foreach_copy_(dst, src)
        for l in range(n):
            w_ih, whh = w_ihs[l], whhs[l]
            H = whh.shape[2]
            last = l == n - 1
            out = _forward_layer(xb, w_ih, whh, lens_dev, T, B, H, last, want_sum=not last)
            y, ybf, gates, csave = out[:4]
            saved += [xb, w_ih, whh, ybf, gates, csave]
            if not last:
                xb = (out[4] if len(out) > 4 else ybf[0, 1:T + 1] + ybf[1, 1:T + 1]).view(T * B, H)
        ctx.save_for_backward(lens_dev, *saved)
        ctx.n = n
        return y.sum(2)

    @staticmethod
    def backward(ctx, dy):
        lens_dev, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        grads = [None] * (4 * ctx.n)
        dy, planes = dy.contiguous(), False
        # opt-in: measured slower than recurrence + GEMM at the bench shape (csrc/lstm.hip,
        # lstm_bwd_dx_kernel)
        fuse_ok = os.environ.get('ASR_LSTM_FUSED_BWD', '0') == '1'
        for l in range(ctx.n - 1, -1, -1):
            xb, w_ih, whh, ybf, gates, csave = saved[6 * l:6 * l + 6]
            _, T2, B, H = ybf.shape
            T, F = T2 - 2, xb.shape[1]
            whhT = whh.transpose(1, 2).contiguous()
            need_dx = l > 0 or ctx.needs_input_grad[0]
            if need_dx and l > 0 and F == H and fuse_ok and _native.lstm_fused_supported(B, H, backward=True):
                # the input gradient comes out of the recurrence (one plane per direction) and
                # goes into the layer below as it is: no [T*B, 8H] x [8H, F] GEMM
                dgb, dy_next = _native.lstm_bidir_bwd_fused(
                    dy, whhT, w_ih.view(2, 4 * H, F).transpose(1, 2).contiguous(), lens_dev,
                    gates, csave, planes=planes)
                planes_next = True
            else:
                dgb = _native.lstm_bidir_bwd(dy, whhT, lens_dev, gates, csave, planes=planes)
                dy_next, planes_next = None, False
                if need_dx:
                    dy_next = _input_gradient(dgb, w_ih, T, B, H, F, bf16_out=l == 0 and ctx.x_bf16)
            dw_ih, dw_hh = _weight_gradients(dgb, xb, ybf, T, B, H, F)
            grads[4 * l:4 * l + 4] = [dw_ih[:4 * H], dw_hh[0], dw_ih[4 * H:], dw_hh[1]]
            dy, planes = dy_next, planes_next
        return (dy if ctx.needs_input_grad[0] else None, None) + tuple(grads)


def bilstm_stack(x, lens, rnns):
    """x [T,B,F] GPU tensor through the nn.LSTM modules `rnns` (bidirectional, bias-free,
    one layer each), directions summed after every layer -> [T,B,H] f32."""
    lens_dev = _native.lens_on(lens, x.device)
    weights = []
    for rnn in rnns:
        weights += [rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.weight_ih_l0_reverse,
                    rnn.weight_hh_l0_reverse]
    return BiLSTMStackFunction.apply(x.contiguous(), lens_dev, *weights)


def bilstm(x, lens, rnn, sum_dirs=False):
    """x [T,B,F] GPU tensor, lens [B] (any int tensor), rnn: nn.LSTM(bidirectional,
    bias=False, 1 layer).  Returns per-direction outputs [T,B,2,H], or their sum
    [T,B,H] with sum_dirs."""
    lens_dev = _native.lens_on(lens, x.device)
    return BiLSTMFunction.apply(
        x.contiguous(), lens_dev, rnn.weight_ih_l0, rnn.weight_hh_l0,
        rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, sum_dirs)
