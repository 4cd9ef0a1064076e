"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/liboracle.so (lattice_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
numpy in, numpy out; no torch, no GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def path_logsumexp(lp, lens, graph_matrices, neg_inf=-1e20):
    """PathLogSumExp.forward (fst_utils.py:403-480) on numpy arrays.
    graph_matrices: the reference's 8-tuple, each [Bg,N,K] / [Bg,N,1].
    Returns dict(logZ [B], grad [T,B,C], alphas [T,B,N], logZ_bwd [B])."""
    lp = _f32(lp)
    T, B, C = lp.shape
    (s_i, l_i, w_i, term, s_o, l_o, w_o, _) = graph_matrices
    s_i, l_i, s_o, l_o = _i32(s_i), _i32(l_i), _i32(s_o), _i32(l_o)
    w_i, w_o, term = _f32(w_i), _f32(w_o), _f32(term)
    Bg, N, Kin = s_i.shape
    Kout = s_o.shape[2]
    lens = _i32(lens)
    logZ = np.zeros(B, np.float32)
    logZb = np.zeros(B, np.float32)
    grad = np.zeros((T, B, C), np.float32)
    alphas = np.zeros((T, B, N), np.float32)
    lib().oracle_path_logsumexp(
        _p(lp), T, B, C, _p(lens), _p(s_i), _p(l_i), _p(w_i), _p(term),
        _p(s_o), _p(l_o), _p(w_o), N, Kin, Kout, Bg,
        ctypes.c_float(neg_inf), _p(logZ), _p(grad), _p(alphas), _p(logZb))
    return dict(logZ=logZ, grad=grad, alphas=alphas, logZ_bwd=logZb)


def path_logsumexp_f64(lp, lens, graph_matrices, neg_inf=-1e20):
    """fp64 arbiter of path_logsumexp: same recurrences and order, double intermediates
    (oracle_path_logsumexp_f64).  Returns dict(logZ [B], grad [T,B,C], logZ_bwd [B]) float64."""
    lp = _f32(lp)
    T, B, C = lp.shape
    (s_i, l_i, w_i, term, s_o, l_o, w_o, _) = graph_matrices
    s_i, l_i, s_o, l_o = _i32(s_i), _i32(l_i), _i32(s_o), _i32(l_o)
    w_i, w_o, term = _f32(w_i), _f32(w_o), _f32(term)
    Bg, N, Kin = s_i.shape
    Kout = s_o.shape[2]
    lens = _i32(lens)
    logZ = np.zeros(B, np.float64)
    logZb = np.zeros(B, np.float64)
    grad = np.zeros((T, B, C), np.float64)
    lib().oracle_path_logsumexp_f64(
        _p(lp), T, B, C, _p(lens), _p(s_i), _p(l_i), _p(w_i), _p(term),
        _p(s_o), _p(l_o), _p(w_o), N, Kin, Kout, Bg,
        ctypes.c_float(neg_inf), _p(logZ), _p(grad), _p(logZb))
    return dict(logZ=logZ, grad=grad, logZ_bwd=logZb)


def path_forward(lp, lens, graph_matrices, neg_inf=-1e20, viterbi=False):
    """path_reduction's alpha-only scan (fst_utils.py:349-397); with
    viterbi=True also the best-path ilabel per frame
    (advanced_decoder.py:546-554).  Returns (score [B], best_il [T,B]|None)."""
    lp = _f32(lp)
    T, B, C = lp.shape
    s_i, l_i, w_i, term = graph_matrices[:4]
    s_i, l_i = _i32(s_i), _i32(l_i)
    w_i, term = _f32(w_i), _f32(term)
    Bg, N, K = s_i.shape
    lens = _i32(lens)
    score = np.zeros(B, np.float32)
    best = np.zeros((T, B), np.int32) if viterbi else None
    lib().oracle_path_forward(
        _p(lp), T, B, C, _p(lens), _p(s_i), _p(l_i), _p(w_i), _p(term),
        N, K, Bg, ctypes.c_float(neg_inf), int(bool(viterbi)), _p(score),
        _p(best) if viterbi else None)
    return score, best


def log_softmax(acts, num_symbols=0, normalize_by_dim=None):
    """get_normalized_acts (ctc_losses.py:29-43) with normalize_logits=True."""
    acts = _f32(acts)
    C = acts.shape[-1]
    R = acts.size // C
    out = np.empty_like(acts)
    mode = 1 if normalize_by_dim else 0
    lib().oracle_log_softmax.argtypes = [
        ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]
    lib().oracle_log_softmax(_p(acts), R, C, int(num_symbols or 1), mode, _p(out))
    return out
