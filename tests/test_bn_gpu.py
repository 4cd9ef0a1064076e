"""Fused BatchNorm2d + Hardtanh (csrc/bnact.hip) against torch's fp32
nn.BatchNorm2d + nn.Hardtanh(0, 20) on the CPU — the pair after each convolution
of the reference's DeepSpeech2 front-end (deep_speech_2.py:60-73)."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,C,H,W', [(3, 32, 50, 17), (5, 4, 7, 11), (2, 32, 334, 11), (2, 6, 9, 5)])
@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('time_major', [False, True])
@pytest.mark.parametrize('channels_last', [False, True])
def test_bn_hardtanh_matches_torch(B, C, H, W, training, time_major, channels_last):
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    torch.manual_seed(B * 100 + C)
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) * 8 + 0.5)            # some channels reach both clamps
        bn.bias.copy_(torch.randn(C) * 4 + 6)
        bn.running_mean.copy_(torch.randn(C) * 0.3)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    act = nn.Hardtanh(0, 20)
    bn.train(training)
    x = torch.randn(B, C, H, W) * 1.5 + 0.2
    dy = torch.randn(B, C, H, W)

    ref_bn = copy.deepcopy(bn)
    xr = x.clone().requires_grad_()
    yr = act(ref_bn(xr))
    yr.backward(dy)

    dev = torch.device('cuda:0')
    bn.to(dev)
    xg = x.to(dev)
    if channels_last:           # the storage MIOpen's NHWC convolutions hand over
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_()
    y = bn_hardtanh(xg, bn, act, time_major=time_major)
    if channels_last and not time_major and C % 4 == 0:     # C = 6: dense-NCHW fallback
        assert y.is_contiguous(memory_format=torch.channels_last)
    dyg = dy.to(dev)
    if time_major:
        assert tuple(y.shape) == (H, B, C, W)
        y.backward(dyg.permute(2, 0, 1, 3).contiguous())
        y_cmp = y.permute(1, 2, 0, 3)
    else:
        y.backward(dyg)
        y_cmp = y
    torch.testing.assert_close(y_cmp.detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-4)
    scale = float(xr.grad.abs().max()) + 1e-6
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 2e-4 * scale
    for got, want in ((bn.weight.grad, ref_bn.weight.grad), (bn.bias.grad, ref_bn.bias.grad)):
        s = float(want.abs().max()) + 1e-6
        assert float((got.cpu() - want).abs().max()) <= 2e-4 * s
    # running statistics and the step counter follow nn.BatchNorm2d
    torch.testing.assert_close(bn.running_mean.cpu(), ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var.cpu(), ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


def test_bn_hardtanh_bf16_output_is_the_rounded_fp32_one():
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    bn = nn.BatchNorm2d(8).to(dev)
    act = nn.Hardtanh(0, 20)
    x = torch.randn(4, 8, 30, 9, device=dev) * 5 + 3
    bn2 = copy.deepcopy(bn)
    y32 = bn_hardtanh(x, bn, act)
    y16 = bn_hardtanh(x, bn2, act, out_bf16=True)
    assert y16.dtype == torch.bfloat16
    assert torch.equal(y16, y32.to(torch.bfloat16))
    # gradients flow from a bf16 dy
    xg = x.clone().requires_grad_()
    bn_hardtanh(xg, bn2, act, out_bf16=True).backward(torch.ones_like(y16))
    assert torch.isfinite(xg.grad).all()


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('channels_last', [False, True])
def test_folded_conv_bias(training, channels_last):
    """conv_bias: the convolution ran bias-free, the fused kernels add its bias on the
    fly and return its gradient (sum of dx; exactly 0 with batch statistics)."""
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    torch.manual_seed(5)
    B, C, H, W = 4, 32, 21, 9
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) * 8 + 0.5)
        bn.bias.copy_(torch.randn(C) * 4 + 6)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    bn.train(training)
    act = nn.Hardtanh(0, 20)
    x = torch.randn(B, C, H, W) * 1.5
    cb = torch.randn(C)
    dy = torch.randn(B, C, H, W)
    ref_bn = copy.deepcopy(bn)
    xr, cbr = x.clone().requires_grad_(), cb.clone().requires_grad_()
    yr = act(ref_bn(xr + cbr[None, :, None, None]))
    yr.backward(dy)

    dev = torch.device('cuda:0')
    bn.to(dev)
    xg = x.to(dev)
    if channels_last:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_()
    cbg = cb.to(dev).requires_grad_()
    y = bn_hardtanh(xg, bn, act, conv_bias=cbg)
    y.backward(dy.to(dev))
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-4)
    scale = float(xr.grad.abs().max()) + 1e-6
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 2e-4 * scale
    # the reference's bias gradient is rounding noise around 0 in training mode
    tol = 2e-4 * (float(cbr.grad.abs().max()) + 1e-6) + (2e-3 if training else 0.0)
    assert float((cbg.grad.cpu() - cbr.grad).abs().max()) <= tol
    torch.testing.assert_close(bn.running_mean.cpu(), ref_bn.running_mean, rtol=1e-5, atol=1e-6)


def test_bf16_channels_last_input():
    """The convolutions hand over bf16 channels-last outputs: statistics, activation and
    gradients must be those of the fp32 computation on the same (bf16-rounded) values."""
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    torch.manual_seed(2)
    dev = torch.device('cuda:0')
    B, C, H, W = 3, 32, 40, 11
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) * 8 + 0.5)
        bn.bias.copy_(torch.randn(C) * 4 + 6)
    act = nn.Hardtanh(0, 20)
    xb = (torch.randn(B, C, H, W) * 1.5).to(torch.bfloat16)
    dy = torch.randn(H, B, C, W)
    ref_bn = copy.deepcopy(bn)
    xr = xb.float().requires_grad_()
    yr = act(ref_bn(xr))
    yr.backward(dy.permute(1, 2, 0, 3))
    bn.to(dev)
    xg = xb.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = bn_hardtanh(xg, bn, act, time_major=True)
    y.backward(dy.to(dev))
    assert xg.grad.dtype == torch.bfloat16
    torch.testing.assert_close(y.permute(1, 2, 0, 3).detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-4)
    scale = float(xr.grad.abs().max())
    assert float((xg.grad.float().cpu() - xr.grad).abs().max()) <= 1e-2 * scale      # bf16 dx
    torch.testing.assert_close(bn.weight.grad.cpu(), ref_bn.weight.grad, rtol=2e-4, atol=1e-3)
    torch.testing.assert_close(bn.running_var.cpu(), ref_bn.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('time_major,out_bf16', [(False, True), (False, False), (True, True), (True, False)])
@pytest.mark.parametrize('B,C,H,W', [(3, 32, 40, 17), (2, 32, 67, 11), (5, 16, 9, 8)])
def test_bf16_input_all_output_forms(B, C, H, W, time_major, out_bf16):
    """bf16 channels-last x through the 16-byte-access kernels (plain channels-last y / dy)
    and the LDS-transposing ones (time-major y / dy), y and dy in fp32 or bf16: the forward
    is the fp32 computation on the bf16-rounded x (rounded once more for bf16 y), the
    gradients those of the fp32 computation with the same (rounded) dy."""
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    torch.manual_seed(B * 31 + W)
    dev = torch.device('cuda:0')
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) * 8 + 0.5)
        bn.bias.copy_(torch.randn(C) * 4 + 6)
    act = nn.Hardtanh(0, 20)
    xb = (torch.randn(B, C, H, W) * 1.5).to(torch.bfloat16)
    dy = torch.randn(B, C, H, W)
    if out_bf16:
        dy = dy.to(torch.bfloat16).float()
    ref_bn = copy.deepcopy(bn)
    xr = xb.float().requires_grad_()
    yr = act(ref_bn(xr))
    yr.backward(dy)
    bn.to(dev)
    xg = xb.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = bn_hardtanh(xg, bn, act, time_major=time_major, out_bf16=out_bf16)
    assert y.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    dyg = dy.to(dev).to(y.dtype)
    if time_major:
        assert tuple(y.shape) == (H, B, C, W)
        y.backward(dyg.permute(2, 0, 1, 3).contiguous())
        y_cmp = y.permute(1, 2, 0, 3)
    else:
        y.backward(dyg.contiguous(memory_format=torch.channels_last))
        y_cmp = y
    if out_bf16:
        assert torch.equal(y_cmp.detach().cpu(), yr.detach().to(torch.bfloat16)) or \
            float((y_cmp.detach().float().cpu() - yr.detach()).abs().max()) <= 2 ** -7 * 20
    else:
        torch.testing.assert_close(y_cmp.detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-4)
    scale = float(xr.grad.abs().max())
    assert float((xg.grad.float().cpu() - xr.grad).abs().max()) <= 1e-2 * scale      # bf16 dx
    torch.testing.assert_close(bn.weight.grad.cpu(), ref_bn.weight.grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(bn.bias.grad.cpu(), ref_bn.bias.grad, rtol=2e-4, atol=2e-3)
