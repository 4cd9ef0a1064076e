"""The CPU oracle (oracle/lattice_oracle.c, oracle/fst_oracle.py) against the
golden vectors produced by the imported reference (tests/golden/make_golden.py).
This is what pins the oracle; the GPU parity tests then compare against it."""
import numpy as np
import pytest

from conftest import golden

LATTICES = ['lattice_mono', 'lattice_bigram_s7', 'lattice_bigram_s49',
            'lattice_den_mono', 'lattice_den_bigram_s7']


@pytest.mark.parametrize('name', LATTICES)
def test_fwbw_matches_reference(oracle_lib, name):
    g = golden(name + '.npz')
    mats = [g['gm%d' % i] for i in range(8)]
    r = oracle_lib.path_logsumexp(g['lp'], g['lens'], mats)
    # reference PathLogSumExp (fst_utils.py:400-488): same fp32 op order
    np.testing.assert_allclose(r['logZ'], g['fwbw_logZ'], rtol=2e-7, atol=2e-6)
    np.testing.assert_allclose(r['grad'], g['fwbw_grad'], rtol=0, atol=5e-6)
    # weighted backward = grad_output * cached grads (fst_utils.py:482-485)
    np.testing.assert_allclose(r['grad'] * g['w'][None, :, None],
                               g['fwbw_grad_w'], rtol=0, atol=5e-6)
    # the reference's own consistency check (fst_utils.py:475-479)
    assert np.abs(r['logZ'] - r['logZ_bwd']).max() < 1e-3
    # rows past the utterance end are zero (fst_utils.py:448)
    for b, l in enumerate(g['lens']):
        assert not r['grad'][l:, b].any()


@pytest.mark.parametrize('name', LATTICES)
def test_fp64_arbiter_agrees_with_reference(oracle_lib, name):
    """oracle_path_logsumexp_f64 (the yardstick of tests/test_lattice_gpu.py::
    assert_posteriors) is the same recurrence: on these short lattices it agrees with the
    imported reference's fp32 outputs to fp32 rounding, and its posterior rows sum to one."""
    g = golden(name + '.npz')
    mats = [g['gm%d' % i] for i in range(8)]
    r = oracle_lib.path_logsumexp_f64(g['lp'], g['lens'], mats)
    assert r['grad'].dtype == np.float64
    np.testing.assert_allclose(r['logZ'], g['fwbw_logZ'], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(r['grad'], g['fwbw_grad'], rtol=0, atol=1e-4)   # the fp32 side's rounding (3e-5 at |logZ| ~ 100)
    np.testing.assert_allclose(r['logZ_bwd'], r['logZ'], rtol=1e-12, atol=1e-9)
    for b, l in enumerate(g['lens']):
        assert not r['grad'][l:, b].any()
        if r['logZ'][b] > -1e19 and name.startswith('lattice_mono'):
            np.testing.assert_allclose(r['grad'][:l, b].sum(-1), 1.0, atol=1e-9)


@pytest.mark.parametrize('name', LATTICES)
def test_forward_and_viterbi_match_reference(oracle_lib, name):
    g = golden(name + '.npz')
    mats = [g['gm%d' % i] for i in range(8)]
    s, _ = oracle_lib.path_forward(g['lp'], g['lens'], mats)
    np.testing.assert_allclose(s, g['autodiff_logZ'], rtol=2e-7, atol=2e-6)
    # the autodiff gradient equals the explicit forward-backward one
    np.testing.assert_allclose(g['autodiff_grad'], g['fwbw_grad'], atol=1e-4)
    v, il = oracle_lib.path_forward(g['lp'], g['lens'], mats, viterbi=True)
    np.testing.assert_array_equal(v, g['viterbi_score'])
    sel = g['viterbi_selidx']
    for b, l in enumerate(g['lens']):            # bit-exact label indices
        np.testing.assert_array_equal(il[:l, b], sel[:l, b])


def test_normalized_acts_match_reference(oracle_lib):
    g = golden('normalized_acts.npz')
    S = int(g['S'])
    np.testing.assert_allclose(oracle_lib.log_softmax(g['acts']), g['zero'], atol=2e-6)
    np.testing.assert_allclose(oracle_lib.log_softmax(g['acts']), g['none_nl'], atol=2e-6)
    np.testing.assert_allclose(oracle_lib.log_softmax(g['acts'], S, 1), g['one'], atol=2e-6)
    np.testing.assert_array_equal(g['none_raw'], g['acts'])


def test_greedy_collapse_matches_reference():
    from oracle import fst_oracle
    g = golden('embedders_greedy.npz')
    S = int(g['S'])
    for tag, order in [('mono', 1), ('bi', 2)]:
        C = S ** order
        blanks = [i for i in range(C) if i % S == 0]
        frames, lens = g['greedy_%s_frames' % tag], g['greedy_%s_lens' % tag]
        flat, dl = g['greedy_%s_flat' % tag], g['greedy_%s_declens' % tag]
        off = 0
        for i in range(len(lens)):
            want = flat[off:off + dl[i]].tolist()
            off += dl[i]
            got = fst_oracle.process_sequence(frames[i], int(lens[i]), blanks, S)
            assert got == want
