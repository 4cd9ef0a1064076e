"""The decode-side caller right after the hot path: reference
`ctc_forward.py:68-131` runs encoder + `decoder.logits`, post-processes the
log-probabilities and writes one float matrix per utterance to a Kaldi archive
that `decode-faster` consumes (`egs/wsj/ctc_kaldi_decode.sh:119-147`).

The post-processing runs on the device on the `[T', B, C]` tensor (the reference
moves it to numpy first); the archive writer emits Kaldi's binary float-matrix
records (`<key> \\0B FM \\4 rows \\4 cols data`), the format
`kaldi_io.BaseFloatMatrixWriter('ark:...')` produces.
"""
import struct

import numpy as np
import torch

EPSILON = 1e-30          # ctc_forward.py:27


def postprocess_logprobs(logprobs, transfer_hash_prob=False, imitate_biphones=False,
                         block_normalize=False, block_marginalize=False,
                         blank_index=0, hash_index=3):
    """ctc_forward.py:96-127 on a `[T', B, C]` tensor (any device).  Branch
    structure as in the reference: `imitate_biphones` tiles the C mono outputs
    C times and renormalises over all C² classes unless `block_normalize` is set;
    otherwise `block_normalize` renormalises inside each context block of a C=S²
    output; otherwise `block_marginalize` averages over contexts and returns S
    classes."""
    lp = logprobs.clone()
    if transfer_hash_prob:                                   # (:97-103)
        blank = lp[:, :, blank_index].exp() + lp[:, :, hash_index].exp() - EPSILON
        lp[:, :, blank_index] = blank.log()
        lp[:, :, hash_index] = float(np.log(EPSILON))
    t, bsz, num_classes = lp.shape
    if imitate_biphones:                                     # (:107-115)
        lp = lp.repeat(1, 1, num_classes)
        if not block_normalize:
            z = lp.exp().sum(dim=2, keepdim=True)
            lp = lp - (z + EPSILON).log()
    elif block_normalize:                                    # (:116-120)
        num_mono = int(np.round(num_classes ** 0.5))
        z = lp.exp().reshape(t, bsz, num_mono, num_mono).sum(dim=3)
        lp = lp - (z.repeat_interleave(num_mono, dim=2) + EPSILON).log()
    elif block_marginalize:                                  # (:121-128)
        num_symbols = int(np.round(num_classes ** 0.5))
        probs = lp.exp().reshape(t, bsz, num_symbols, num_symbols).sum(dim=2) / num_symbols
        lp = probs.log()
        assert not bool(torch.isnan(lp).any())
    return lp


class KaldiFloatMatrixWriter(object):
    """Binary Kaldi archive of float32 matrices; `writer[key] = matrix`."""

    def __init__(self, wspecifier):
        path = wspecifier[4:] if wspecifier.startswith('ark:') else wspecifier
        self.f = open(path, 'wb')

    def __setitem__(self, key, mat):
        mat = np.ascontiguousarray(mat, dtype=np.float32)
        assert mat.ndim == 2
        self.f.write(key.encode('ascii') + b' \0BFM ')
        self.f.write(b'\4' + struct.pack('<i', mat.shape[0]) + b'\4' + struct.pack('<i', mat.shape[1]))
        self.f.write(mat.tobytes())

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_kaldi_float_matrices(path):
    """Reader for the records above (tests / round trips)."""
    out = {}
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        sp = data.index(b' ', pos)
        key = data[pos:sp].decode('ascii')
        assert data[sp + 1:sp + 6] == b'\0BFM ' and data[sp + 6:sp + 7] == b'\4'
        rows = struct.unpack('<i', data[sp + 7:sp + 11])[0]
        assert data[sp + 11:sp + 12] == b'\4'
        cols = struct.unpack('<i', data[sp + 12:sp + 16])[0]
        n = rows * cols * 4
        out[key] = np.frombuffer(data[sp + 16:sp + 16 + n], np.float32).reshape(rows, cols).copy()
        pos = sp + 16 + n
    return out


@torch.no_grad()
def forward_batch(model, features, feature_lens, speakers, uttids, owriter, **post):
    """One batch of ctc_forward.py:80-131: encoder, `decoder.logits`, the
    post-processing above, then one archive record per utterance (sorted by
    utterance id, cut to its encoded length)."""
    encoded, encoded_lens = model.encoder(features, feature_lens, speakers)
    logprobs = model.decoder.logits(encoded, encoded_lens)          # [T', B, C]
    logprobs = postprocess_logprobs(logprobs, **post).cpu().numpy()
    for i in np.argsort(uttids):
        owriter[uttids[i]] = logprobs[:int(encoded_lens[i]), i, :]
    return logprobs
