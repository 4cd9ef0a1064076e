"""The class projection of the decoders for wide alphabets (C = 2401): split-bf16 MFMA
products (att_speech/modules/decoders/advanced_decoder.py, reference advanced_decoder.py:79-223
computes F.linear in fp32) against an fp64 product of the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(7,), (1000, 320), (333, 2401), (5, 3, 17)])
def test_split_bf16_halves(shape):
    from att_speech import _native
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * torch.exp(torch.randn(*shape, generator=g) * 4)).to(d)
    x.view(-1)[0] = 0.0
    hi, lo = _native.split_bf16(x)
    assert torch.equal(hi, x.to(torch.bfloat16))                       # round to nearest even
    assert torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    err = (hi.double() + lo.double() - x.double()).abs()
    assert (err <= x.double().abs() * 2.0 ** -16).all()
    if len(shape) == 2:                                                # strided destinations: the K-concatenated operand
        rows, K = shape
        a = torch.zeros(rows, 3 * K, dtype=torch.bfloat16, device=d)
        _native.split_bf16(x, a[:, :K], a[:, 2 * K:])
        assert torch.equal(a[:, :K], hi) and torch.equal(a[:, 2 * K:], lo) and not a[:, K:2 * K].any()


@pytest.mark.parametrize('rows,F,C,bias', [(4096, 320, 2401, True), (8192, 320, 2401, False), (4100, 64, 300, True)])
def test_wide_projection_matches_fp64(rows, F, C, bias, monkeypatch):
    """forward and the three gradients within a few 2^-16 of the operands' magnitudes — an
    order of magnitude closer to fp64 than a plain bf16 product, and not further from it than
    4x the fp32 library product's own distance plus that margin"""
    from att_speech.modules.decoders import advanced_decoder as ad
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, F, generator=g).to(d).requires_grad_(True)
    layer = torch.nn.Linear(F, C, bias=bias).to(d)
    layer.class_weight_bias = lambda: (layer.weight, layer.bias)
    dy = torch.randn(rows, C, generator=g).to(d)

    def run(split):
        monkeypatch.setenv('ASR_PROJ_SPLIT', '1' if split else '0')
        for p in (x, layer.weight, layer.bias):
            if p is not None:
                p.grad = None
        y = ad.project_frames(layer, x)
        y.backward(dy)
        return [y.detach().double()] + [p.grad.double() for p in (x, layer.weight, layer.bias) if p is not None]

    got, ref32 = run(True), run(False)
    xd, wd = x.detach().double(), layer.weight.detach().double()
    want = [xd @ wd.t() + (layer.bias.detach().double() if bias else 0), dy.double() @ wd, dy.double().t() @ xd]
    if bias:
        want.append(dy.double().sum(0))
    for name, a, r, w in zip(('y', 'dx', 'dW', 'db'), got, ref32, want):
        scale = w.abs().max().item()
        e_split, e_f32 = (a - w).abs().max().item(), (r - w).abs().max().item()
        assert e_split <= 4 * e_f32 + 4e-5 * scale, (name, e_split, e_f32, scale)


@pytest.mark.parametrize('T,B,F,C,bias', [(40, 128, 320, 2401, True), (33, 130, 320, 2401, False), (50, 100, 64, 300, True)])
def test_fused_projection_normalise_shift(T, B, F, C, bias):
    """one autograd node (the logits' gradient leaves the normalisation's backward as bf16
    halves + column sums) == projection node followed by the normalise/shift node"""
    from att_speech import _native
    from att_speech.modules.decoders import advanced_decoder as ad
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(T, B, F, generator=g).to(d).requires_grad_(True)
    layer = torch.nn.Linear(F, C, bias=bias).to(d)
    layer.class_weight_bias = lambda: (layer.weight, layer.bias)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g).sort(descending=True)[0].to(d, torch.int32)
    dsh = torch.rand(T, B, C, generator=g).to(d)          # posteriors-like: non-negative
    params = [p for p in (x, layer.weight, layer.bias) if p is not None]

    def grads(fn):
        for p in params:
            p.grad = None
        shifted, msum = fn()
        shifted.backward(dsh)
        return [shifted.detach(), msum.detach()] + [p.grad.clone() for p in params]

    fused = grads(lambda: ad.project_normalise_shift(layer, x, lens))
    plain = grads(lambda: ad._NormaliseShift.apply(ad.project_frames(layer, x), lens))
    for name, a, b in zip(('shifted', 'max_sum', 'dx', 'dW', 'db'), fused, plain):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 3e-5 * scale + 1e-6, (name, (a - b).abs().max().item(), scale)
