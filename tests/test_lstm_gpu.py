"""Native BiLSTM recurrence (csrc/lstm.hip, bf16 MFMA operands / fp32 state)
against torch's fp32 nn.LSTM on a packed batch on the CPU — the op the
reference's BatchRNN runs (encoder_utils.py:78,100,112-117)."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(x, lens, rnn, dy):
    x = x.clone().requires_grad_()
    packed = nn.utils.rnn.pack_padded_sequence(x, lens)
    y, _ = rnn(packed)
    y, _ = nn.utils.rnn.pad_packed_sequence(y, total_length=x.size(0))
    y.backward(dy)
    return y.detach(), x.grad, [p.grad.clone() for p in rnn.parameters()]


@pytest.mark.parametrize('T,B,F,H,lens', [
    (37, 5, 48, 64, [37, 30, 30, 11, 1]),
    (60, 40, 352, 320, None),          # first encoder layer shape, 2 batch tiles
    (12, 33, 320, 320, None),
])
def test_bilstm_matches_packed_torch_lstm(T, B, F, H, lens):
    from att_speech.modules.encoders.native_lstm import bilstm
    torch.manual_seed(T * 1000 + B)
    if lens is None:
        lens = sorted(np.random.RandomState(B).randint(1, T + 1, size=B).tolist(), reverse=True)
        lens[0] = T
    lens_t = torch.tensor(lens)
    rnn = nn.LSTM(F, H, bidirectional=True, bias=False)
    x = torch.randn(T, B, F)
    dy = torch.randn(T, B, 2 * H)
    mask = (torch.arange(T)[:, None] < lens_t[None, :]).float()[:, :, None]
    dy = dy * mask
    y_ref, dx_ref, dw_ref = _ref(x, lens_t, rnn, dy)

    dev = torch.device('cuda:0')
    rnn_g = nn.LSTM(F, H, bidirectional=True, bias=False)
    rnn_g.load_state_dict(rnn.state_dict())
    rnn_g.to(dev)
    xg = x.to(dev).requires_grad_()
    y = bilstm(xg, lens_t, rnn_g).view(T, B, 2 * H)
    y.backward(dy.to(dev))

    def close(a, b, what, rtol):
        a, b = a.detach().cpu(), b.detach().cpu()
        err = float((a - b).abs().max())
        scale = float(b.abs().max()) + 1e-6
        assert err <= rtol * scale, (what, err, scale)

    # bf16 operands (8-bit mantissa), fp32 accumulation and state
    close(y, y_ref, 'y', 3e-2)
    assert not y.detach().cpu()[mask.expand_as(y_ref) == 0].any()   # zeros on padding
    close(xg.grad, dx_ref, 'dx', 5e-2)
    for p, w in zip(rnn_g.parameters(), dw_ref):
        close(p.grad, w, 'dw', 5e-2)
