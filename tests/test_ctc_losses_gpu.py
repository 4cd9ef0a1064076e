"""GPU parity of the dense-CTC loss front-ends: `ctc_raw_loss_batch` /
`ctc_raw_loss` and `CTCDecoderAdvanced(ctc_loss_fn='ctc_raw_loss_batch')` on device tensors
against the outputs of the reference's own dense transition-matrix CTC
(ctc_losses.py:67-166, 327-390, 563-609; tests/golden/make_golden.py::golden_dense_bicontext,
golden_lattice_mono).  The product evaluates them with the sparse lattice kernel."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device('cuda:0')


def _cat(labels, llens):
    return torch.cat([torch.from_numpy(labels[b, :llens[b]]).long() for b in range(len(llens))])


@pytest.mark.parametrize('fn_name', ['ctc_raw_loss_batch', 'ctc_raw_loss'])
def test_bicontext_dense_golden(fn_name):
    from att_speech.modules import ctc_losses as L
    fn = getattr(L, fn_name)
    d = golden('dense_bicontext.npz')
    S = int(d['S'])
    lens, llens = torch.from_numpy(d['lens']), torch.from_numpy(d['label_lens']).long()
    acts = torch.from_numpy(d['acts']).to(dev()).requires_grad_()
    loss = fn(acts, _cat(d['labels'], d['label_lens']), lens, llens, num_symbols=S,
              context_order=2, normalize_by_dim=1, eval_repeats_in_context=True)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), d['loss_rep_ctx'], rtol=1e-5)
    loss.sum().backward()
    np.testing.assert_allclose(acts.grad.cpu().numpy(), d['grad_acts_rep_ctx'], atol=5e-5)
    # default repeat handling (by symbol): equal on repeat-free transcripts, refused otherwise
    with pytest.raises(NotImplementedError):
        fn(acts.detach(), _cat(d['labels'], d['label_lens']), lens, llens, num_symbols=S,
           context_order=2, normalize_by_dim=1)
    T1 = int(d['lens'][1])
    a1 = torch.from_numpy(d['acts'][:T1, 1:]).to(dev()).requires_grad_()
    l1 = fn(a1, _cat(d['labels'][1:], d['label_lens'][1:]), lens[1:], llens[1:], num_symbols=S,
            context_order=2, normalize_by_dim=1)
    np.testing.assert_allclose(l1.detach().cpu().numpy(), d['loss_rep_sym'][1:], rtol=1e-5)
    l1.sum().backward()
    np.testing.assert_allclose(a1.grad.cpu().numpy(), d['grad_acts_rep_sym'][:T1, 1:], atol=5e-5)


def test_mono_dense_golden():
    from att_speech.modules import ctc_losses as L
    g = golden('lattice_mono.npz')
    lens, llens = torch.from_numpy(g['lens']), torch.from_numpy(g['label_lens']).long()
    acts = torch.from_numpy(g['lp']).to(dev()).requires_grad_()
    loss = L.ctc_raw_loss_batch(acts, _cat(g['labels'], g['label_lens']), lens, llens,
                                num_symbols=49, context_order=1)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g['dense_loss'], rtol=1e-5)
    loss.sum().backward()
    np.testing.assert_allclose(acts.grad.cpu().numpy(), g['dense_grad_acts'], atol=5e-5)


def test_decoder_with_raw_loss_fn():
    """CTCDecoderAdvanced(ctc_loss_fn='ctc_raw_loss_batch') (advanced_decoder.py:238,292-297):
    the ten positional arguments reach the loss; identity projection so the golden applies."""
    from att_speech.modules.decoders import CTCDecoderAdvanced
    d = golden('dense_bicontext.npz')
    S = int(d['S'])
    C = S * S
    T, B = d['acts'].shape[:2]
    dec = CTCDecoderAdvanced({'features': torch.zeros(T, 2, C)}, C, context_order=2,
                             normalize_by_dim=1, ctc_loss_fn='ctc_raw_loss_batch',
                             vocabulary=['s%d' % i for i in range(S)])
    with torch.no_grad():
        lin = dec.fc[0].module[0]
        lin.weight.copy_(torch.eye(C))
        lin.bias.zero_()
    dec = dec.to(dev())
    # positional slot ten of ctc_raw_loss_batch is eval_repeats_in_context; the decoder puts
    # other_data_in_batch (an empty dict here, i.e. falsy) there, as the reference does
    T1 = int(d['lens'][1])
    enc = torch.from_numpy(d['acts'][:T1, 1:]).to(dev())
    out = dec(enc, torch.from_numpy(d['lens'][1:]), torch.from_numpy(d['labels'][1:]),
              torch.from_numpy(d['label_lens'][1:]))
    np.testing.assert_allclose(float(out['loss']), float(d['loss_rep_sym'][1:].sum()), rtol=1e-5)


@pytest.mark.parametrize('order,dim', [(2, 1), (3, 1), (3, 2)])
def test_get_normalized_acts_over_any_context_axis(order, dim):
    """get_normalized_acts(normalize_by_dim = d) = log_softmax over axis d + 2 of the
    [T, B, S, ..., S] view (reference ctc_losses.py:29-43, restated here with torch on the CPU):
    values and gradient, for the last axis (the group kernel directly) and a strided one."""
    from att_speech.modules.ctc_losses import get_normalized_acts
    S, T, B = 5, 7, 3
    g = torch.Generator().manual_seed(11)
    acts = torch.randn(T, B, S ** order, generator=g) * 3
    w = torch.randn(T, B, S ** order, generator=g)
    ref_in = acts.clone().requires_grad_(True)
    ref = torch.nn.functional.log_softmax(ref_in.view(T, B, *(S,) * order), dim + 2).view(T, B, -1)
    (ref * w).sum().backward()
    x = acts.to(dev()).requires_grad_(True)
    y = get_normalized_acts(x, None, S, order, dim)
    (y * w.to(dev())).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_in.grad.numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(IndexError):
        get_normalized_acts(x, None, S, order, order)
