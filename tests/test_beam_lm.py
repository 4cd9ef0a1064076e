"""LM-fused searches: BeamSearchLM / RescoreSearchLM / GraphSearch and
the LM bag functions against outputs of the reference's own code on a toy LM
(tests/golden/beam_lm.npz; make_golden.py runs reference beam_search.py:185-648 and
fst_utils.py:23-188 through an in-memory 2-to-3 shim), plus an independent brute-force
path enumeration and the LmFst file round trips.  Host-side code: no GPU needed."""
import itertools

import numpy as np
import pytest
import torch

from conftest import golden


def toy_lm(g):
    from att_speech.lm_fst import LmFst, SymbolTable
    syms = SymbolTable([(0, '<eps>'), (1, '<spc>'), (2, 'a'), (3, 'b'), (4, 'c')])
    return LmFst(6, 0, g['lm_src'], g['lm_dst'], g['lm_il'], g['lm_il'], g['lm_w'],
                 g['lm_final'], syms, syms)


def flat_bags(bags):
    return np.array([(l, k, v) for l, d in enumerate(bags) for k, v in sorted(d.items())],
                    np.float64)


def test_bag_functions_match_reference_and_brute_force():
    from att_speech import fst_utils as P
    g = golden('beam_lm.npz')
    lm = toy_lm(g)
    nodes = {0: 0.0, 3: 0.4, 5: 1.1}
    for name, logp in (('log', True), ('min', False)):
        np.testing.assert_allclose(flat_bags(P.expand_all(lm, 7, dict(nodes), logp)),
                                   g['bags_%s' % name], rtol=1e-12)
        e = P.expand_epsilon(lm, {4: 0.1, 5: 0.2, 1: 0.3}, logp)
        np.testing.assert_allclose(np.array(sorted(e.items())), g['eps_%s' % name], rtol=1e-12)
        spc_bag = {int(k): v for k, v in g['spc_bag']}
        got = [P.score_nodes(lm, dict(nodes), False, logp), P.score_nodes(lm, dict(nodes), True, logp),
               P.score_nodes(lm, spc_bag, True, logp, '<spc>')]
        np.testing.assert_allclose(got, g['score_%s' % name], rtol=1e-12)
        for l in range(5):
            assert P.expand(lm, dict(nodes), l, logp) == P.expand_all(lm, 7, dict(nodes), logp)[l]
    # independent check: enumerate every path "one arc of label l, then epsilon arcs"
    arcs = list(zip(g['lm_src'].tolist(), g['lm_dst'].tolist(), g['lm_il'].tolist(),
                    g['lm_w'].tolist()))
    for l in (1, 2, 3, 4):
        paths = {}

        def walk(s, c):
            paths.setdefault(s, []).append(c)
            for a, b, il, w in arcs:
                if a == s and il == 0:
                    walk(b, c + w)
        for s0, c0 in nodes.items():
            for a, b, il, w in arcs:
                if a == s0 and il == l:
                    walk(b, c0 + w)
        want = {s: -np.logaddexp.reduce(-np.array(c)) for s, c in paths.items()}
        got = P.expand(lm, dict(nodes), l, True)
        assert set(got) == set(want)
        for s in want:
            assert abs(got[s] - want[s]) < 1e-12
    assert P.reduce_weights([], True) == float('inf')
    with pytest.raises(IndexError):
        P.expand_all(lm, 3, dict(nodes), True)          # LM label outside the classes


def test_epsilon_cycle_is_an_error_only_when_reachable():
    from att_speech import fst_utils as P
    from att_speech.lm_fst import LmFst
    lm = LmFst(4, 0, [0, 1, 2, 3], [1, 2, 1, 3], [1, 0, 0, 1], [1, 0, 0, 1],
               [0.5, 0.1, 0.2, 0.3], np.zeros(4))
    assert lm.eps_rank() is None
    assert P.expand(lm, {3: 0.0}, 1, True) == {3: 0.3}
    with pytest.raises(ValueError):
        P.expand(lm, {0: 0.0}, 1, True)


def test_lm_fst_file_round_trips(tmp_path):
    from att_speech.lm_fst import LmFst
    lm = toy_lm(golden('beam_lm.npz'))
    lm.write(str(tmp_path / 'lm.fst'))
    back = LmFst.read(str(tmp_path / 'lm.fst'))
    assert back.start() == lm.start() and back.num_states() == 6
    assert list(back.input_symbols()) == list(lm.input_symbols())
    for s in range(6):
        a, b = list(lm.arcs(s)), list(back.arcs(s))
        assert [(x.ilabel, x.nextstate) for x in a] == [(x.ilabel, x.nextstate) for x in b]
        np.testing.assert_allclose([x.weight for x in a], [x.weight for x in b], rtol=1e-6)
        assert [x.ilabel for x in a] == sorted(x.ilabel for x in a)     # ilabel-sorted
        assert (back.final(s) == lm.final(s)) or abs(back.final(s) - lm.final(s)) < 1e-6
    # AT&T text format with a symbol table
    (tmp_path / 'syms.txt').write_text('<eps> 0\n<spc> 1\na 2\nb 3\nc 4\n')
    (tmp_path / 'lm.txt').write_text('0 1 a a 0.5\n0 0 <spc> <spc>\n1 0 <eps> <eps> 0.25\n1 1.5\n0\n')
    t = LmFst.read_text(str(tmp_path / 'lm.txt'), str(tmp_path / 'syms.txt'))
    assert t.start() == 0 and t.final(1) == 1.5 and t.final(0) == 0.0
    assert [(a.ilabel, a.nextstate, a.weight) for a in t.arcs(0)] == [(1, 0, 0.0), (2, 1, 0.5)]
    assert t.input_symbols().find('<spc>') == 1


def _np(x):
    return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def drive(bs, g, tag, n, nb, device='cpu'):
    for i in range(n):
        l, m = bs.step(torch.from_numpy(g['logits'][i][:, :nb]).clone().to(device),
                       att_weights=torch.from_numpy(g['att'][i][:, :nb]).clone().to(device))
        np.testing.assert_array_equal(l.cpu().numpy(), g[tag + '_letters'][i])
        np.testing.assert_array_equal(m.cpu().numpy(), g[tag + '_maps'][i])
        np.testing.assert_allclose(bs.scores.cpu().numpy(), g[tag + '_scores'][i], rtol=1e-5)
    assert len(bs.finished) == int(g[tag + '_nfinished'])
    np.testing.assert_allclose([float(f[0]) for f in bs.finished], g[tag + '_finished_scores'],
                               rtol=1e-5)
    assert [int(f[2]) for f in bs.finished] == g[tag + '_finished_beams'].tolist()
    fl = [_np(f[1]) for f in bs.finished]
    np.testing.assert_array_equal(np.concatenate(fl) if fl else np.zeros(0, np.int64),
                                  g[tag + '_finished_flat'])
    assert [len(f) for f in fl] == g[tag + '_finished_lens'].tolist()
    np.testing.assert_array_equal(_np(bs.best_finished[0]), g[tag + '_best'])
    np.testing.assert_allclose(float(bs.best_finished_scores[0]), float(g[tag + '_best_score']),
                               rtol=1e-5)
    for k, v in bs.best_finished_scores_elements.items():
        np.testing.assert_allclose(v, g[tag + '_el_' + k], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(bs.estimations.cpu().numpy(), g[tag + '_estimations'])
    st = np.array([(b, k, v) for b, d in enumerate(bs.fst_states) for k, v in sorted(d.items())],
                  np.float64).reshape(-1, 3)
    np.testing.assert_allclose(st, g[tag + '_fst_states'], rtol=1e-10)
    if bs.coverage is not None:
        np.testing.assert_allclose(bs.coverage.cpu().numpy(), g[tag + '_coverage'], rtol=1e-6)
    return bs


@pytest.mark.parametrize('device', ['cpu', pytest.param('cuda:0', marks=pytest.mark.gpu)])
def test_lm_fused_searches_match_reference(device):
    """on host tensors and (`-m gpu`) with the decoder's logits / alignments
    living on the MI355X, as AttentionDecoderTCN.decode hands them over"""
    from att_speech.modules.beam_search import BeamSearchLM, GraphSearch, RescoreSearchLM
    g = golden('beam_lm.npz')
    lm, mapping = toy_lm(g), g['mapping'].tolist()
    C, beam, steps = 7, 4, g['logits'].shape[0]
    dev = torch.device(device)
    drive(BeamSearchLM(lm, 0.5, mapping, 0.3, 0.1, 0.2, 1, beam, dev, C, 0.6,
                       keep_eos_score=False), g, 'lm', steps, beam, device)
    drive(BeamSearchLM(lm, 0.8, mapping, 0.2, 0.1, 0.0, 1, beam, dev, C, 0.0,
                       keep_eos_score=True), g, 'lmk', steps, beam, device)
    r = drive(RescoreSearchLM(g['sentence'].tolist(), lm, 0.5, mapping, 0.3, 0.1, 0.2, 1, 1, dev,
                              C, 0.6, keep_eos_score=False), g, 'rs', 6, 1, device)
    np.testing.assert_allclose(r.attentions.cpu().numpy(), g['rs_attentions'], rtol=1e-6)

    def hash_dec(decoded, hs=2):
        return hash(tuple([-1] * (hs - len(decoded)) + decoded[-hs:].tolist()))
    gs = drive(GraphSearch(hash_dec, 0.3, lm, 0.5, mapping, 0.3, 0.1, 0.2, 1, beam, dev, C, 0.6,
                           keep_eos_score=False), g, 'gs', steps, beam, device)
    G = gs.get_graph()[0]
    V = np.array([[v[0], -1 if v[1] == '<sos>' else v[1], int(bool(v[4]))] for v in G['V']],
                 np.int64)
    np.testing.assert_array_equal(V, g['gs_V'])
    np.testing.assert_allclose([v[2] for v in G['V']], g['gs_V_scores'], rtol=1e-5)
    E = np.array([[e[0], e[1], int(e[2] == 'merged')] for e in G['E']], np.int64).reshape(-1, 3)
    np.testing.assert_array_equal(E, g['gs_E'])


def test_tcn_decoder_selects_the_lm_searches():
    """AttentionDecoderTCN(lm_file=...) builds the alphabet mapping of tcn.py:306-327 and
    decodes with the LM-fused searches (smoke: runs, returns the reference's result keys)."""
    from att_speech.modules.tcn import AttentionDecoderTCN
    from att_speech.modules.beam_search import BeamSearchLM, GraphSearch
    lm = toy_lm(golden('beam_lm.npz'))
    vocab = ['<pad>', '<unk>', ' ', 'a', 'b', 'c']
    torch.manual_seed(0)
    enc = torch.randn(14, 1, 16)
    kw = dict(tcn_hidden_size=24, att_hidden_size=8, dropout_p=0.0, kernel_size=3,
              dilation_sizes=[1, 2], beam_size=3, length_normalization=0.6, vocabulary=vocab,
              lm_file=lm, lm_weight=0.5, coverage_weight=0.1, coverage_tau=0.1,
              min_attention_pos=0.1)
    for graph in (False, True):
        dec = AttentionDecoderTCN({'features': torch.zeros(enc.shape)}, 6,
                                  use_graph_search=graph, **kw).eval()
        assert dec.alphabet_mapping == [1, 1, 1, 2, 3, 4, 1]
        dec.TRANSCRIPTION_LEN_GUARD = 8
        with torch.no_grad():
            res = dec.decode(enc, torch.tensor([14]))
        assert isinstance(res['beam_search'], GraphSearch if graph else BeamSearchLM)
        assert set(res) >= {'decoded', 'decoded_scores', 'loss', 'coverage', 'graph'}
        assert res['beam_search'].estimations.shape == (3, 8) or res['beam_search'].has_finished()
        assert (res['graph'] is not None) == graph
