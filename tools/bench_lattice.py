"""Kernel micro-benchmark for the lattice scans (development tool; bench.py is
the contract benchmark).  Prints per-launch time and algorithmic GB/s
(SURVEY.md §8d byte formula)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
sys.path.insert(0, ROOT)

from att_speech import _native, fst_utils   # noqa: E402


def algorithmic_bytes(lens, C, n_states, n_arcs):
    """4*T'_b*(3C + 2N_b) + 2*E_b*12 + 4*N_b summed over utterances."""
    lens = np.asarray(lens, np.int64)
    return int((4 * lens * (3 * C + 2 * n_states) + 24 * n_arcs + 4 * n_states).sum())


def make(order, B, T, kind, seed=1234):
    rng = np.random.default_rng(seed)
    S = 49
    C = S ** order
    lens = np.full(B, T, np.int32)
    gg = fst_utils.CTCGraphGen(context_order=order, num_symbols=S)
    if kind == 'num':
        llens = np.array([100 - 2 * (b % 16) for b in range(B)])
        labs = rng.integers(2, 49, size=(B, 100))
        mats = gg.get_training_matrices_batch(labs, llens)
        n_states = 2 * llens + 1
        n_arcs = (mats[2] > -1e19).sum((1, 2)).numpy()
    else:
        mats = gg.get_decoding_matrices()
        n_states = np.full(B, mats[0].shape[1])
        n_arcs = np.full(B, int((mats[2] > -1e19).sum()))
    return lens, mats, C, n_states, n_arcs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='mono_num,bi_num,mono_den')
    ap.add_argument('--B', type=int, default=512)
    ap.add_argument('--T', type=int, default=334)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for case in a.cases.split(','):
        order = 1 if case.startswith('mono') else 2
        kind = 'num' if case.endswith('num') else 'den'
        B = a.B
        lens, mats, C, n_states, n_arcs = make(order, B, a.T, kind)
        g = _native.Graph(mats, dev)
        x = torch.randn(a.T, B, C, device=dev)
        lp = _native.log_softmax_fwd(x, C)
        tl = torch.from_numpy(lens).to(dev)
        for _ in range(3):
            _native.lattice_fwbw(lp, tl, g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            _native.lattice_fwbw(lp, tl, g)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        by = algorithmic_bytes(lens, C, n_states, n_arcs)
        print("%-9s B=%d T=%d C=%d N=%d K=%d: fwbw %.3f ms  alg %.1f MB  %.1f GB/s (%.1f%% of 8 TB/s)" % (
            case, B, a.T, C, g.N, g.Kin, ms, by / 1e6, by / ms / 1e6, by / ms / 1e6 / 80.0))
        e0.record()
        for _ in range(a.iters):
            _native.lattice_forward(lp, tl, g, viterbi=True, want_path=True)
        e1.record()
        torch.cuda.synchronize()
        print("%-9s viterbi+path %.3f ms" % (case, e0.elapsed_time(e1) / a.iters))
        if kind == 'den':
            gg = _native.GroupedGraph(mats.grouped, dev)
            for _ in range(2):
                _native.grouped_fwbw(lp, tl, gg)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                _native.grouped_fwbw(lp, tl, gg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print("%-9s GROUPED fwbw %.3f ms  (%.1f GB/s algorithmic)" % (case, ms, by / ms / 1e6))
            e0.record()
            for _ in range(a.iters):
                _native.grouped_forward(lp, tl, gg, viterbi=True, want_path=True)
            e1.record()
            torch.cuda.synchronize()
            print("%-9s GROUPED viterbi+path %.3f ms" % (case, e0.elapsed_time(e1) / a.iters))
        e0.record()
        for _ in range(a.iters):
            _native.log_softmax_fwd(x, C)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print("%-9s log_softmax fwd %.3f ms  %.1f GB/s" % (case, ms, 2 * x.numel() * 4 / ms / 1e6))


if __name__ == '__main__':
    main()
