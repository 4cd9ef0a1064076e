"""TCN attention decoder + plain BeamSearch against the
outputs of the reference's own classes (tests/golden/tcn_beam.npz, see
make_golden.py: tcn.py imported unchanged, BeamSearch executed from the py3-clean
first 182 lines of beam_search.py)."""
import warnings

import numpy as np
import pytest
import torch

from conftest import golden

warnings.filterwarnings('ignore')

KW = dict(tcn_hidden_size=24, att_hidden_size=8, dropout_p=0.0, kernel_size=3,
          dilation_sizes=[1, 2], beam_size=3, length_normalization=0.6,
          attention_temperature=1.25, tcn_layers_per_block=2)


def build(g, device):
    from att_speech.modules.tcn import AttentionDecoderTCN
    S = int(g['S'])
    enc = torch.from_numpy(g['enc'])
    dec = AttentionDecoderTCN({'features': torch.zeros(enc.shape)}, S, **KW)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd_')}
    assert set(sd) == set(dec.state_dict())          # checkpoint-compatible keys
    dec.load_state_dict(sd)
    return dec.eval().to(device), enc.to(device)


def check(device, tol):
    g = golden('tcn_beam.npz')
    dec, enc = build(g, device)
    lens = torch.from_numpy(g['lens'])
    out = dec(enc, lens, torch.from_numpy(g['texts']), torch.from_numpy(g['text_lens']),
              return_att_weights=True)
    np.testing.assert_allclose(float(out['loss']), float(g['fwd_loss']), rtol=tol)
    np.testing.assert_allclose(out['logits'].detach().cpu().numpy(), g['fwd_logits'],
                               atol=tol * 10)
    np.testing.assert_allclose(torch.stack(out['attweights']).detach().cpu().numpy(),
                               g['fwd_att'], atol=tol)
    dec.TRANSCRIPTION_LEN_GUARD = 12
    with torch.no_grad():
        res = dec.decode(enc, lens, return_attention=True)
    assert set(res) >= {'decoded', 'decoded_scores', 'loss', 'coverage', 'graph', 'beam_search'}
    got = [[int(c) for c in (d.tolist() if hasattr(d, 'tolist') else d)] for d in res['decoded']]
    off, want = 0, []
    for n in g['dec_lens']:
        want.append(g['dec_flat'][off:off + n].tolist())
        off += n
    assert got == want                                          # label indices bit-exact
    np.testing.assert_allclose(np.array(res['decoded_scores']['acoustic']), g['dec_scores'],
                               rtol=tol * 10)
    bs = res['beam_search']
    assert bs.finished_count == g['dec_finished_count'].tolist()
    np.testing.assert_array_equal(bs.estimations.cpu().numpy(), g['dec_final_estimations'])
    np.testing.assert_allclose(bs.scores.cpu().numpy(), g['dec_final_beam_scores'], rtol=tol * 10)
    np.testing.assert_allclose(torch.cat(res['logits']).cpu().numpy(), g['dec_step_logits'],
                               atol=tol * 20)


def test_tcn_decoder_and_beam_search_match_reference_cpu():
    check(torch.device('cpu'), 1e-5)


@pytest.mark.gpu
def test_tcn_decoder_and_beam_search_match_reference_gpu():
    check(torch.device('cuda:0'), 2e-4)
