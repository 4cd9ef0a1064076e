"""The encoder's 32 -> 32 channel 7x7 convolution (reference deep_speech_2.py:60-73,
Conv2d(32, 32, (7, 7), stride (3, 1))) as one autograd function over the hand-written MFMA
kernels of csrc/conv.hip (include/asr_amd.h: asr_conv7x7c32_{fwd,bwd_data,wgrad}_bf16):
channels-last bf16 activations, fp32 accumulation, fp32 weight gradient, bias-free (the bias
lives in the fused BatchNorm kernels).  The module keeps its nn.Conv2d parameters (state_dict
keys `conv.3.{weight,bias}` unchanged)."""
import torch

from att_speech import _native


def supported(conv, x):
    """the shapes csrc/conv.hip is built for; anything else stays on torch / MIOpen"""
    return (x.is_cuda and isinstance(conv, torch.nn.Conv2d) and conv.in_channels == 32
            and conv.out_channels == 32 and tuple(conv.kernel_size) == (7, 7)
            and tuple(conv.stride) == (3, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and x.dim() == 4 and x.size(2) >= 7 and 7 <= x.size(3) <= 48)


class Conv7x7C32Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, weight)
        y, sums = _native.conv7x7c32_fwd(x, weight.detach(), 3, want_sums=True)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums):
        x, weight = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _native.conv7x7c32_bwd_data(dy, weight.detach(), x.size(2), x.size(3), 3)
        if ctx.needs_input_grad[1]:
            dw = _native.conv7x7c32_wgrad(x, dy, 3)
        return dx, dw


def first_supported(conv, x):
    """Conv2d(cin, 32, (7, 7), stride (1, 2), padding (6, 0)) on [B, cin, T, F] features: one
    channel (the synthetic 40-dim benchmark features), or the WSJ recipes' three (81 mel bins:
    static, delta, delta-delta) with an even output width"""
    return (x.is_cuda and isinstance(conv, torch.nn.Conv2d) and conv.in_channels in (1, 3)
            and conv.out_channels == 32 and tuple(conv.kernel_size) == (7, 7)
            and tuple(conv.stride) == (1, 2) and tuple(conv.padding) == (6, 0)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and x.dim() == 4 and x.size(1) == conv.in_channels
            and _native.conv1_supported(x.size(3), conv.in_channels)
            and not x.requires_grad)        # the kernel pair computes no feature gradient


class Conv1Function(torch.autograd.Function):
    """features [B, T, F] or [B, T, F, cin] f32 -> [B, 32, T + 6, Fo] channels-last bf16; the
    features need no gradient, the weight gradient is asr_conv1{,c}_7x7s2_wgrad"""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.float().contiguous()
        ctx.save_for_backward(x, weight)
        y, sums = _native.conv1_fwd(x, weight.detach(), want_sums=True)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums):
        x, weight = ctx.saved_tensors
        return None, (_native.conv1_wgrad(x, dy) if ctx.needs_input_grad[1] else None)


def conv1(features, conv):
    """features logical [B, cin, T, F] (any strides; the reference's [B, T, F, cin] batches
    arrive as a permuted view, deep_speech_2.py:127) -> (conv(features) without the bias, its
    channel sums [2, 32] f64 for the following BatchNorm)"""
    if features.size(1) == 1:
        return Conv1Function.apply(features[:, 0], conv.weight)
    return Conv1Function.apply(features.permute(0, 2, 3, 1), conv.weight)


def conv7x7c32(x, conv):
    """(y = conv(x) without the bias: logical [B, 32, Ho, Wo] bf16, channels-last memory;
    channel sums [2, 32] f64 of y for the following BatchNorm)"""
    return Conv7x7C32Function.apply(x, conv.weight)
