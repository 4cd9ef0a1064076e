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


@pytest.mark.gpu
def test_native_decode_step_at_recipe_dims_matches_cpu_mirror():
    """lattice_decoding/tcn.yaml dimensions (TCN 384, attention 64, 2 blocks x 2 layers, k = 3,
    dilations [1, 2], 49 symbols + EOS, beam 10) with seeded random weights: the MI355X decode
    loop (TCN.last_step products, asr_tcn_attention_step_f32, asr_beam_step_f32, flag polled
    every 8 steps) against the host mirror of the reference's loop (full TCN, BeamSearch with a
    read-back per step) on the CPU: label indices, finished counts and hypotheses bit-exact."""
    from att_speech.modules.tcn import AttentionDecoderTCN
    torch.manual_seed(4242)
    S, T, B, E = 49, 90, 5, 320
    kw = dict(tcn_hidden_size=384, att_hidden_size=64, dropout_p=0.0, kernel_size=3,
              dilation_sizes=[1, 2], beam_size=10, length_normalization=0.6,
              attention_temperature=1.25, tcn_layers_per_block=2)
    dec = AttentionDecoderTCN({'features': torch.zeros(T, B, E)}, S, **kw).eval()
    with torch.no_grad():
        for prm in dec.parameters():
            prm.add_(torch.randn_like(prm) * 0.05)
        dec.output_to_logits.bias[S] += 1.5            # EOS competitive: hypotheses finish
    enc = torch.randn(T, B, E)
    lens = torch.tensor([90, 81, 77, 60, 41])
    dec.TRANSCRIPTION_LEN_GUARD = 40
    with torch.no_grad():
        want = dec.decode(enc, lens)
        dev = torch.device('cuda:0')
        dec_g = dec.to(dev)
        assert dec_g._native_decode_ok(enc.to(dev))
        got = dec_g.decode(enc.to(dev), lens)
    from att_speech.modules.beam_search import DeviceBeamSearch
    assert isinstance(got['beam_search'], DeviceBeamSearch)
    as_list = lambda d: [int(c) for c in (d.tolist() if hasattr(d, 'tolist') else d)]
    assert [as_list(d) for d in got['decoded']] == [as_list(d) for d in want['decoded']]
    assert any(len(as_list(d)) > 0 for d in want['decoded'])
    wb, gb = want['beam_search'], got['beam_search']
    assert gb.finished_count == wb.finished_count
    np.testing.assert_array_equal(gb.estimations.cpu().numpy(), wb.estimations.numpy())
    np.testing.assert_allclose(gb.scores.cpu().numpy(), wb.scores.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(np.array(got['decoded_scores']['acoustic']),
                               np.array(want['decoded_scores']['acoustic']), rtol=2e-4)
