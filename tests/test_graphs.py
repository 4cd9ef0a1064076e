"""Host logic: the product's closed-form graph generator
(att_speech.fst_utils.CTCGraphGen) against the oracle's literal mini-OpenFst
restatement, and the loss identities that validate both (pywrapfst is absent, so
graph construction is 'parity unpinned' at the OpenFst boundary — see
oracle/fst_oracle.py)."""
import numpy as np
import pytest
import torch

from conftest import golden

CASES = [(1, 49, {}), (2, 7, {}), (2, 7, dict(use_contextual_blanks=True)),
         (2, 7, dict(allow_nonblank_selfloops=False)), (2, 49, {})]


@pytest.mark.parametrize('order,S,kw', CASES)
def test_closed_form_equals_composition(order, S, kw):
    from att_speech import fst_utils as P
    from oracle import fst_oracle as O
    rng = np.random.default_rng(order * 100 + S)
    B, Lmax = 7, 9
    lens = np.array([9, 8, 7, 5, 3, 1, 0])
    labs = rng.integers(1, min(S, 5), size=(B, Lmax))
    labs[0, :4] = [2, 2, 2, 3]                       # repeats incl. a triple
    pg = P.CTCGraphGen(context_order=order, num_symbols=S, graph_build_args=kw)
    og = O.CTCGraphGen(S, order, graph_build_args=kw)
    pm = pg.get_training_matrices_batch(labs, lens)
    om = og.get_training_matrices_batch(labs, lens)
    assert len(pm) == len(om) == 8
    for a, b in zip(pm, om):
        assert a.dtype == (torch.int64 if b.dtype == np.int64 else torch.float32)
        np.testing.assert_array_equal(a.numpy(), b)
    for a, b in zip(pg.get_decoding_matrices(), og.get_decoding_matrices()):
        np.testing.assert_array_equal(a.numpy(), b)
    # single-utterance form + reference batching (fst_utils.py:491-521)
    singles = [pg.get_training_matrices(labs[i, :lens[i]]) for i in range(B)]
    for a, b in zip(P.batch_training_graph_matrices(singles), pm):
        np.testing.assert_array_equal(a.numpy(), b.numpy())


def test_bigram_ids_are_reduced_modulo_symbols():
    from att_speech import fst_utils as P
    S = 7
    labs = np.array([[3, 4, 4, 2]])
    big = np.array([[0 * S + 3, 3 * S + 4, 4 * S + 4, 4 * S + 2]])
    pg = P.CTCGraphGen(context_order=2, num_symbols=S)
    for a, b in zip(pg.get_training_matrices_batch(labs, [4]),
                    pg.get_training_matrices_batch(big, [4])):
        np.testing.assert_array_equal(a.numpy(), b.numpy())


def test_decoding_read_out_matches_transducer_walk():
    from att_speech import fst_utils as P
    from oracle import fst_oracle as O
    rng = np.random.default_rng(5)
    for order, S in [(1, 9), (2, 5)]:
        pg = P.CTCGraphGen(context_order=order, num_symbols=S)
        og = O.CTCGraphGen(S, order)
        nxt, _ = pg.decoding_fst.transition_tables()
        for _ in range(20):
            s, ils = 0, []
            for _t in range(40):
                il = rng.choice(np.where(nxt[s] >= 0)[0])
                ils.append(il)
                s = nxt[s, il]
            assert (pg.decoding_fst.read_out(ils) ==
                    O.read_out_olabels(og.decoding_fst, ils))


def test_mono_lattice_equals_torch_ctc_loss(oracle_lib):
    """identity PathLogSumExp == F.ctc_loss on product graphs."""
    from att_speech import fst_utils as P
    rng = np.random.default_rng(11)
    S, T, B, Lmax = 49, 60, 6, 12
    lens = np.array([60, 55, 51, 40, 33, 30], np.int32)
    llens = np.array([12, 10, 9, 7, 3, 1])
    labs = rng.integers(2, 49, size=(B, Lmax))
    labs[1, 3] = labs[1, 4]
    lp = torch.log_softmax(torch.from_numpy(
        rng.standard_normal((T, B, S)).astype(np.float32)), -1)
    pg = P.CTCGraphGen(context_order=1, num_symbols=S)
    mats = [m.numpy() for m in pg.get_training_matrices_batch(labs, llens)]
    r = oracle_lib.path_logsumexp(lp.numpy(), lens, mats)
    lpt = lp.clone().requires_grad_()
    want = torch.nn.functional.ctc_loss(
        lpt, torch.from_numpy(labs), torch.from_numpy(lens).long(),
        torch.from_numpy(llens), reduction='none')
    np.testing.assert_allclose(-r['logZ'], want.detach().numpy(), rtol=1e-5)
    # gradients agree at the logits level (F.ctc_loss folds the softmax in)
    logits = lp.clone().requires_grad_()
    torch.nn.functional.ctc_loss(
        torch.log_softmax(logits, -1), torch.from_numpy(labs),
        torch.from_numpy(lens).long(), torch.from_numpy(llens),
        reduction='sum').backward()
    gs = -torch.from_numpy(r['grad'])
    gl = gs - lp.exp() * gs.sum(-1, keepdim=True)
    np.testing.assert_allclose(gl.numpy(), logits.grad.numpy(), atol=5e-5)


def test_mono_and_bigram_lattices_equal_reference_dense_ctc(oracle_lib):
    """sparse closed-form graphs vs the reference's DENSE transition-matrix CTC
    (ctc_losses.py:67-166,327-390) — an oracle independent of OpenFst."""
    from att_speech import fst_utils as P
    g = golden('lattice_mono.npz')
    np.testing.assert_allclose(-g['fwbw_logZ'], g['dense_loss'], rtol=2e-6)
    pg = P.CTCGraphGen(context_order=1, num_symbols=49)
    mats = [m.numpy() for m in pg.get_training_matrices_batch(g['labels'], g['label_lens'])]
    r = oracle_lib.path_logsumexp(g['lp'], g['lens'], mats)
    np.testing.assert_allclose(-r['logZ'], g['dense_loss'], rtol=2e-6)

    d = golden('dense_bicontext.npz')
    S = int(d['S'])
    pg = P.CTCGraphGen(context_order=2, num_symbols=S,
                       graph_build_args=dict(use_contextual_blanks=True))
    mats = [m.numpy() for m in pg.get_training_matrices_batch(d['labels'], d['label_lens'])]
    r = oracle_lib.path_logsumexp(d['log_probs'], d['lens'], mats)
    # the FST evaluates repeats in context (fst_utils.py:808): equals the dense
    # model with eval_repeats_in_context=True on every utterance ...
    np.testing.assert_allclose(-r['logZ'], d['loss_rep_ctx'], rtol=2e-6)
    # ... and the default dense model wherever no label repeats (utt 0 repeats)
    np.testing.assert_allclose(-r['logZ'][1:], d['loss_rep_sym'][1:], rtol=2e-6)
    assert abs(-r['logZ'][0] - d['loss_rep_sym'][0]) > 1e-2
    # logits-level gradient
    gs = -torch.from_numpy(r['grad'])
    lp = torch.from_numpy(d['log_probs'])
    p = lp.exp().view(lp.shape[0], lp.shape[1], S, S)
    gl = gs.view_as(p) - p * gs.view_as(p).sum(-1, keepdim=True)
    np.testing.assert_allclose(gl.reshape(lp.shape).numpy(),
                               d['grad_acts_rep_ctx'], atol=5e-5)
