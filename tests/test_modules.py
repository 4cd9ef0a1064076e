"""Host-side mirror of the reference's module surface (no GPU compute):
constructor kwargs, state_dict keys, embedder arithmetic against the imported
reference's outputs, bug-compatible greedy collapse."""
import numpy as np
import torch

from conftest import golden

VOCAB = [str(i) for i in range(49)]
ENC = dict(class_name='att_speech.modules.encoders.DeepSpeech2',
           conv_kernel_sizes=[[7, 7], [7, 7]], conv_strides=[[1, 2], [3, 1]],
           rnn_hidden_size=320, rnn_nb_layers=4, rnn_normalization='none')
DEC_MONO = dict(class_name='att_speech.modules.decoders.advanced_decoder.FSTDecoder',
                denominator_red='none', normalize_by_dim=0,
                graph_generator=dict(class_name='CTCGraphGen', context_order=1))
DEC_CDE = dict(class_name='att_speech.modules.decoders.advanced_decoder.FSTDecoder',
               graph_generator=dict(class_name='CTCGraphGen', context_order=2),
               embedder='NGramLinear',
               embedder_kwargs=dict(bias_only_for_dim=1, num_layers=3,
                                    embedding_combination_method='concat',
                                    tied_embeddings=False))


def sample_batch(B=2, T=100, F=40, ch=1):
    return {'features': torch.randn(B, T, F, ch),
            'features_lengths': torch.tensor([T - 8 * b for b in range(B)]),
            'spkids': ['s'] * B, 'ivectors': None}


def test_model_surface_and_state_dict_keys():
    from att_speech.models import SpeechModel
    m = SpeechModel(ENC, DEC_MONO, sample_batch(), 49, VOCAB)
    # synthetic 40-dim input: conv out 32x11
    assert m.encoder.rnn_input_size == 352
    assert sum(p.numel() for p in m.encoder.parameters()) == 6687456
    keys = set(m.state_dict().keys())
    want = {'encoder.conv.0.weight', 'encoder.conv.0.bias', 'encoder.conv.3.weight',
            'encoder.conv.3.bias', 'decoder.fc.0.module.0.weight',
            'decoder.fc.0.module.0.bias'}
    for i in (1, 4):
        for k in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
            want.add('encoder.conv.%d.batch_norm.%s' % (i, k))
    for l in range(4):
        for k in ('weight_ih_l0', 'weight_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse'):
            want.add('encoder.rnns.%d.rnn.%s' % (l, k))
    assert keys == want
    m2 = SpeechModel(ENC, DEC_CDE, sample_batch(), 2401, VOCAB)
    dk = {k for k in m2.state_dict() if k.startswith('decoder.')}
    assert dk == {'decoder.fc.0.module.0.bias', 'decoder.fc.0.module.0.embedding.weight'} | {
        'decoder.fc.0.module.0.weight_computer.%d.%s' % (i, k)
        for i in (0, 2, 4) for k in ('weight', 'bias')}
    # WSJ shape: 81 mel+energy x 3 channels -> conv out 32x32 -> rnn input 1024
    m3 = SpeechModel(ENC, DEC_MONO, sample_batch(F=81, ch=3), 49, VOCAB)
    assert m3.encoder.rnn_input_size == 1024
    # encoder output contract: (T', B, 320), lens = ceil(len/3)
    enc, lens = m.encoder(torch.randn(2, 100, 40, 1), torch.tensor([100, 92]), None)
    assert tuple(enc.shape) == (34, 2, 320) and lens.tolist() == [34, 31]


def test_registry_resolves_reference_dotted_paths():
    from att_speech import utils
    assert utils.get_class('att_speech.models.SpeechModel').__name__ == 'SpeechModel'
    assert utils.get_class('DeepSpeech2', 'att_speech.modules.encoders').__name__ == 'DeepSpeech2'
    assert utils.get_class('CTCGraphGen', 'att_speech.fst_utils').__name__ == 'CTCGraphGen'
    g = utils.contruct_from_kwargs(dict(class_name='CTCGraphGen', context_order=1),
                                   'att_speech.fst_utils', {'num_symbols': 49, 'num_classes': 49})
    assert g.num_classes == 49


def test_embedders_match_reference_outputs():
    from att_speech import fst_utils
    from att_speech.modules.decoders import LutLinear, NGramLinear
    g = golden('embedders_greedy.npz')
    S = int(g['S'])
    x = torch.from_numpy(g['x'])
    n2c = fst_utils.make_full_ngram_table(2, S, S * S)[2]
    lut = LutLinear(x.shape[1], S, n2c).eval()
    lut.load_state_dict({'weight': torch.from_numpy(g['lut_weight']),
                         'bias': torch.from_numpy(g['lut_bias'])})
    np.testing.assert_allclose(lut(x).detach().numpy(), g['lut_y'], atol=1e-5)
    lut_t = LutLinear(x.shape[1], S, n2c, tie_blanks=True).eval()
    lut_t.load_state_dict(lut.state_dict())
    np.testing.assert_allclose(lut_t(x).detach().numpy(), g['lut_tied_y'], atol=1e-5)
    ng = NGramLinear(x.shape[1], S, n2c, bias_only_for_dim=1, num_layers=3,
                     embedding_combination_method='concat', tied_embeddings=False).eval()
    ng.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files
                        if k.startswith('ng_') and k != 'ng_y'})
    np.testing.assert_allclose(ng(x).detach().numpy(), g['ng_y'], atol=1e-5)
    ng2 = NGramLinear(x.shape[1], S, n2c, embedding_combination_method='sum',
                      num_layers=0, tied_embeddings=True).eval()
    ng2.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith('ng2_') and k != 'ng2_y'})
    np.testing.assert_allclose(ng2(x).detach().numpy(), g['ng2_y'], atol=1e-5)


def test_greedy_collapse_is_bug_compatible():
    from att_speech.modules.decoders import CTCDecoderAdvanced
    g = golden('embedders_greedy.npz')
    S = int(g['S'])
    for tag, order in [('mono', 1), ('bi', 2)]:
        C = S ** order
        dec = CTCDecoderAdvanced(
            {'features': torch.zeros(4, 1, 8)}, C, context_order=order,
            ctc_loss_fn='ctc_fst_loss', vocabulary=[str(i) for i in range(S)])
        frames, lens = g['greedy_%s_frames' % tag], g['greedy_%s_lens' % tag]
        flat, dl = g['greedy_%s_flat' % tag], g['greedy_%s_declens' % tag]
        got = dec.process_sequences(torch.from_numpy(frames), torch.from_numpy(lens))
        off = 0
        for i in range(len(lens)):
            assert got[i] == flat[off:off + dl[i]].tolist()
            off += dl[i]
