"""Hand-written 7x7 32->32 convolution kernels (csrc/conv.hip) against fp32 nn.Conv2d on the
CPU — the op DeepSpeech2's second convolution runs (deep_speech_2.py:60-73)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (the last case has more (utterance, row block) items than the persistent grid has workgroups)
CASES = [(2, 58, 17, 3), (3, 31, 13, 1), (1, 1006, 17, 3), (5, 40, 23, 3), (2, 7, 7, 1), (24, 1006, 17, 3)]


def _inputs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(32, 32, 7, 7, generator=g) * 0.05
    return x, w


@pytest.mark.parametrize('B,H,W,sh', CASES)
def test_conv_forward_matches_conv2d(B, H, W, sh):
    from att_speech import _native
    x, w = _inputs(B, H, W, B * 100 + H)
    dev = torch.device('cuda:0')
    xb = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, sums = _native.conv7x7c32_fwd(xb, w.to(dev), sh, want_sums=True)
    torch.cuda.synchronize()
    # the epilogue's channel statistics are those of the bf16 outputs themselves
    yd = y.double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), yd.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sums[1].cpu().numpy(), (yd * yd).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5)
    # reference on the SAME bf16-rounded operands, fp32 arithmetic
    want = F.conv2d(xb.float().cpu(), w.to(torch.bfloat16).float(), None, (sh, 1))
    assert tuple(y.shape) == tuple(want.shape)
    got = y.float().cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-2 * scale     # bf16 output rounding: 2^-8
    # the fp32 accumulation itself: compare before the output rounding where it is exact
    rel = float(((got - want).abs() / (want.abs() + 1e-2 * scale)).max())
    assert rel < 2e-2


# (24 x 31 row blocks: more items than workgroups, the grid's second half runs the reversed wave roles)
@pytest.mark.parametrize('B,H,W', [(2, 58, 17), (1, 1006, 17), (3, 40, 23), (2, 9, 7), (2, 62, 17), (24, 1006, 17),
                                   (2, 100, 48), (3, 50, 8)])
def test_conv_input_gradient_matches_conv2d(B, H, W):
    from att_speech import _native
    x, w = _inputs(B, H, W, B * 7 + H)
    Ho, Wo = (H - 7) // 3 + 1, W - 6
    g = torch.Generator().manual_seed(H)
    dy = torch.randn(B, 32, Ho, Wo, generator=g)
    dev = torch.device('cuda:0')
    dyb = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dx = _native.conv7x7c32_bwd_data(dyb, w.to(dev), H, W, 3)
    torch.cuda.synchronize()
    want = torch.nn.grad.conv2d_input((B, 32, H, W), w.to(torch.bfloat16).float(),
                                      dyb.float().cpu(), stride=(3, 1))
    got = dx.float().cpu()
    assert tuple(got.shape) == (B, 32, H, W)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-2 * scale


@pytest.mark.parametrize('B,H,W', [(2, 58, 17), (1, 1006, 17), (3, 40, 23), (2, 9, 7), (70, 31, 17),
                                   (2, 64, 38)])
def test_conv_weight_gradient_matches_conv2d(B, H, W):
    from att_speech import _native
    x, w = _inputs(B, H, W, B * 11 + H)
    Ho, Wo = (H - 7) // 3 + 1, W - 6
    g = torch.Generator().manual_seed(H + 1)
    dy = torch.randn(B, 32, Ho, Wo, generator=g)
    dev = torch.device('cuda:0')
    xb = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dyb = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = _native.conv7x7c32_wgrad(xb, dyb, 3)
    torch.cuda.synchronize()
    want = torch.nn.grad.conv2d_weight(xb.float().cpu(), (32, 32, 7, 7), dyb.float().cpu(), stride=(3, 1))
    got = dw.cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-3 * scale      # fp32 accumulation of exact products


# (F = 53 / 81: the weight-gradient kernel's other prefetch-register sizes; 40 x 1000 x 40: more
#  chunks than persistent workgroups)
@pytest.mark.parametrize('B,T,F', [(2, 50, 40), (1, 1000, 40), (3, 33, 81), (2, 5, 9), (2, 30, 53), (40, 1000, 40)])
def test_first_convolution_forward_and_weight_gradient(B, T, F):
    """Conv2d(1, 32, 7x7, stride (1, 2), padding (6, 0)) on the raw features"""
    from att_speech import _native
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, T, F, generator=g)
    w = torch.randn(32, 1, 7, 7, generator=g) * 0.1
    dev = torch.device('cuda:0')
    y, sums = _native.conv1_fwd(x.to(dev), w.to(dev), want_sums=True)
    yd = y.double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), yd.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sums[1].cpu().numpy(), (yd * yd).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5)
    xr = x.to(torch.bfloat16).float()[:, None]                       # operands as the kernel rounds them
    want = F_conv(xr, w.to(torch.bfloat16).float())
    got = y.float().cpu()
    assert tuple(got.shape) == tuple(want.shape)
    assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max())
    dy = torch.randn(want.shape, generator=g)
    dyb = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = _native.conv1_wgrad(x.to(dev), dyb)
    torch.cuda.synchronize()
    want_dw = torch.nn.grad.conv2d_weight(xr, (32, 1, 7, 7), dyb.float().cpu(), stride=(1, 2),
                                          padding=(6, 0))
    assert float((dw.cpu() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())


# (the WSJ recipes' features: 81 mel bins x 3 channels -> Fo = 38; 49 x 3 -> Fo = 22; more chunks
#  than persistent workgroups; a last chunk of fewer than 8 rows)
@pytest.mark.parametrize('B,T,F', [(2, 50, 81), (1, 333, 81), (3, 7, 49), (300, 31, 81), (5, 1000, 81)])
def test_first_convolution_three_input_channels(B, T, F):
    """Conv2d(3, 32, 7x7, stride (1, 2), padding (6, 0)) on [B, T, F, 3] features — the layout
    the reference's batches have before deep_speech_2.py:127 permutes them"""
    from att_speech import _native
    g = torch.Generator().manual_seed(B * 31 + T + F)
    x = torch.randn(B, T, F, 3, generator=g)
    w = torch.randn(32, 3, 7, 7, generator=g) * 0.1
    dev = torch.device('cuda:0')
    assert _native.conv1_supported(F, 3)
    y, sums = _native.conv1_fwd(x.to(dev), w.to(dev), want_sums=True)
    yd = y.double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), yd.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sums[1].cpu().numpy(), (yd * yd).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5)
    xr = x.to(torch.bfloat16).float().permute(0, 3, 1, 2)           # operands as the kernel rounds them
    want = F_conv(xr, w.to(torch.bfloat16).float())
    got = y.float().cpu()
    assert tuple(got.shape) == tuple(want.shape)
    assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max())
    dy = torch.randn(want.shape, generator=g)
    dyb = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = _native.conv1_wgrad(x.to(dev), dyb)
    torch.cuda.synchronize()
    want_dw = torch.nn.grad.conv2d_weight(xr, (32, 3, 7, 7), dyb.float().cpu(), stride=(1, 2),
                                          padding=(6, 0))
    assert float((dw.cpu() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())


def test_first_convolution_three_channels_module_path():
    """native_conv.conv1 takes the reference's permuted [B, 3, T, F] view without a copy"""
    from att_speech.modules.encoders import native_conv
    dev = torch.device('cuda:0')
    conv = torch.nn.Conv2d(3, 32, (7, 7), stride=(1, 2), padding=(6, 0)).to(dev)
    feats = torch.randn(2, 60, 81, 3, device=dev)
    x = feats.permute(0, 3, 1, 2)
    assert native_conv.first_supported(conv, x)
    y, sums = native_conv.conv1(x, conv)
    want = F.conv2d(x.to(torch.bfloat16).float(), conv.weight.detach().to(torch.bfloat16).float(), None, (1, 2), (6, 0))
    assert float((y.float() - want).abs().max()) <= 1e-2 * float(want.abs().max())
    y.float().square().sum().backward()
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
    assert not native_conv.first_supported(conv, torch.randn(2, 3, 60, 80, device=dev))     # odd Fo


def F_conv(x, w):
    return F.conv2d(x, w, None, (1, 2), (6, 0))


def test_first_conv_with_feature_gradients_stays_on_torch():
    """asr_conv1_7x7s2_* computes no gradient for the features: features that require one
     must not take the native path, which would silently drop it."""
    from att_speech.modules.encoders import native_conv
    dev = torch.device('cuda:0')
    conv = torch.nn.Conv2d(1, 32, (7, 7), stride=(1, 2), padding=(6, 0)).to(dev)
    x = torch.randn(2, 1, 50, 40, device=dev)
    assert native_conv.first_supported(conv, x)
    assert not native_conv.first_supported(conv, x.clone().requires_grad_())
