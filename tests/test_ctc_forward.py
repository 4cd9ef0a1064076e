"""Decode-side post-processing + Kaldi archive records: against the
reference's own ctc_forward.py:96-128 block (tests/golden/ctc_forward.npz — make_golden.py
executes that block of the reference file in memory on seeded inputs) and against a numpy
restatement of it."""
import numpy as np
import pytest
import torch

from conftest import golden


def _check_against_reference_block(device, rtol, atol):
    from att_speech.ctc_forward import postprocess_logprobs
    g = golden('ctc_forward.npz')
    assert abs(float(g['EPSILON']) - 1e-30) < 1e-36
    for i in range(int(g['n'])):
        th, ib, bn, bm = [bool(v) for v in g['flags_%d' % i]]
        got = postprocess_logprobs(torch.from_numpy(g['in_%d' % i]).to(device),
                                   transfer_hash_prob=th, imitate_biphones=ib,
                                   block_normalize=bn, block_marginalize=bm).cpu().numpy()
        want = g['out_%d' % i]
        assert got.shape == want.shape, (i, got.shape, want.shape)
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg='combo %d' % i)


def test_postprocess_matches_reference_block_cpu():
    _check_against_reference_block('cpu', 1e-5, 1e-6)


@pytest.mark.gpu
def test_postprocess_matches_reference_block_gpu():
    """the same on device tensors (the decode path keeps the log-probs on the GPU)"""
    _check_against_reference_block('cuda:0', 1e-5, 2e-6)

EPS = 1e-30


def numpy_reference(logprobs, transfer_hash_prob, imitate_biphones, block_normalize,
                    block_marginalize):
    logprobs = logprobs.copy()
    if transfer_hash_prob:                                   # ctc_forward.py:97-103
        blank = np.exp(logprobs[:, :, 0]) + np.exp(logprobs[:, :, 3]) - EPS
        logprobs[:, :, 0] = np.log(blank)
        logprobs[:, :, 3] = np.log(EPS)
    t, bsz, c = logprobs.shape
    if imitate_biphones:                                     # :107-115
        logprobs = np.tile(logprobs, (1, 1, c))
        if not block_normalize:
            z = np.exp(logprobs).sum(axis=2, keepdims=True)
            logprobs -= np.log(z + EPS)
    elif block_normalize:                                    # :116-120
        m = int(np.round(c ** 0.5))
        z = np.exp(logprobs).reshape(t, bsz, m, m).sum(axis=3).repeat(m, axis=2)
        logprobs -= np.log(z + EPS)
    elif block_marginalize:                                  # :121-128
        m = int(np.round(c ** 0.5))
        probs = np.exp(logprobs).reshape(t, bsz, m, m).sum(axis=2) / m
        logprobs = np.log(probs)
    return logprobs


@pytest.mark.parametrize('flags', [
    dict(), dict(transfer_hash_prob=True), dict(imitate_biphones=True),
    dict(imitate_biphones=True, block_normalize=True), dict(block_normalize=True),
    dict(block_marginalize=True), dict(transfer_hash_prob=True, block_marginalize=True)])
def test_postprocess_matches_reference_arithmetic(flags):
    from att_speech.ctc_forward import postprocess_logprobs
    full = dict(transfer_hash_prob=False, imitate_biphones=False, block_normalize=False,
                block_marginalize=False)
    full.update(flags)
    c = 7 if full['imitate_biphones'] else 49
    g = torch.Generator().manual_seed(3)
    lp = torch.log_softmax(torch.randn(11, 3, c, generator=g) * 3, -1)
    got = postprocess_logprobs(lp, **full).numpy()
    want = numpy_reference(lp.numpy(), **full)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_kaldi_archive_round_trip(tmp_path):
    from att_speech.ctc_forward import KaldiFloatMatrixWriter, read_kaldi_float_matrices
    rng = np.random.RandomState(0)
    mats = {'utt_b': rng.randn(5, 49).astype(np.float32), 'utt_a': rng.randn(1, 3).astype(np.float32)}
    path = str(tmp_path / 'logits.ark')
    with KaldiFloatMatrixWriter('ark:' + path) as w:
        for k in sorted(mats):
            w[k] = mats[k]
    raw = open(path, 'rb').read()
    assert raw.startswith(b'utt_a \0BFM \4\1\0\0\0\4\3\0\0\0')          # Kaldi binary FM header
    back = read_kaldi_float_matrices(path)
    assert sorted(back) == ['utt_a', 'utt_b']
    for k in mats:
        np.testing.assert_array_equal(back[k], mats[k])
