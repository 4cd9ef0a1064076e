"""att_speech.fst_utils — MI355X-native counterpart of the reference module of
the same dotted name (reference: att_speech/fst_utils.py).

Kept surface: path_reduction, path_logsumexp / PathLogSumExp,
batch_training_graph_matrices, make_full_ngram_table, CTCGraphGen (context
orders 1 and 2) with get_training_matrices(_batch) / get_decoding_matrices, the
padded-adjacency matrix format of fst_to_matrices.

What is different by design:
  * the lattice arithmetic runs in hand-written gfx950 kernels behind the C ABI
    (include/asr_amd.h) — there is no torch/CPU fallback;
  * graphs are not built with OpenFst: the CTC training lattice
    compose(decoding_fst, chain(labels)) and the decoding graphs have a closed
    form which is generated directly, vectorised over the
    batch, in the same state numbering a breadth-first OpenFst composition
    yields ([blank0, l1, blank1, l2, ...]).
"""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch

from att_speech import _native

NEG_INF = -1e20


# ----------------------------------------------------------------------------
# generic arc list -> padded adjacency matrices (reference fst_to_matrices,
# fst_utils.py:222-294), vectorised with numpy
# ----------------------------------------------------------------------------

def _arcs_to_matrices(n_states, own, other, ilabel, weight, nc_weight):
    """For each `own` state list its arcs sorted by (other, ilabel, weight)
    (fst_utils.py:285), padded to the max degree with (0, 0, nc_weight)."""
    own = np.asarray(own, np.int64)
    other = np.asarray(other, np.int64)
    ilabel = np.asarray(ilabel, np.int64)
    weight = np.asarray(weight, np.float32)
    order = np.lexsort((weight, ilabel, other, own))
    own, other, ilabel, weight = own[order], other[order], ilabel[order], weight[order]
    counts = np.bincount(own, minlength=n_states)
    k = int(counts.max()) if len(own) else 0
    k = max(k, 1)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    slot = np.arange(len(own)) - starts[own]
    states = np.zeros((n_states, k), np.int64)
    ilabels = np.zeros((n_states, k), np.int64)
    weights = np.full((n_states, k), nc_weight, np.float32)
    states[own, slot] = other
    ilabels[own, slot] = ilabel
    weights[own, slot] = weight
    return states, ilabels, weights


def arcs_to_graph_matrices(n_states, src, dst, ilabel, weight, final_weight,
                           nc_weight=NEG_INF, for_forward_only=False):
    """Arc list (src -> dst consuming 0-based ilabel with log-weight) + final
    log-weights -> the reference's 4 (in-edge) or 8 (in- then out-edge)
    matrices, as torch CPU tensors (int64 / float32 like the reference)."""
    term = np.asarray(final_weight, np.float32).reshape(n_states, 1)
    mats = list(_arcs_to_matrices(n_states, dst, src, ilabel, weight, nc_weight))
    mats.append(term)
    if not for_forward_only:
        mats += list(_arcs_to_matrices(n_states, src, dst, ilabel, weight, nc_weight))
        mats.append(term.copy())
    return tuple(torch.from_numpy(m) for m in mats)


def in_to_out_edge_matrices(graph_matrices, nc_weight=NEG_INF):
    """Given the 4 in-edge matrices [Bg,N,K] build the 4 out-edge ones (used
    when a caller hands 4 matrices to a reduction that needs a backward pass:
    the reference differentiates the alpha scan with autograd,
    fst_utils.py:384-394; the kernels run an explicit beta scan instead)."""
    s_i, l_i, w_i, term = [np.asarray(m.cpu()) for m in graph_matrices[:4]]
    bg, n, k = s_i.shape
    outs = []
    kmax = 1
    for g in range(bg):
        valid = w_i[g] > nc_weight * 0.5
        dst = np.repeat(np.arange(n), k).reshape(n, k)[valid]
        o = _arcs_to_matrices(n, s_i[g][valid], dst, l_i[g][valid], w_i[g][valid],
                              nc_weight)
        outs.append(o)
        kmax = max(kmax, o[0].shape[1])
    so = np.zeros((bg, n, kmax), np.int64)
    lo = np.zeros((bg, n, kmax), np.int64)
    wo = np.full((bg, n, kmax), nc_weight, np.float32)
    for g, (a, b, c) in enumerate(outs):
        so[g, :, :a.shape[1]] = a
        lo[g, :, :a.shape[1]] = b
        wo[g, :, :a.shape[1]] = c
    return (torch.from_numpy(so), torch.from_numpy(lo), torch.from_numpy(wo),
            torch.from_numpy(term.copy()))


class GraphMatrices(list):
    """The reference's list of 4 / 8 padded graph tensors, optionally carrying
    the group-factored description of the same graph (`grouped`, see
    DecodingTransducer.grouped_structure) so that path_reduction can use the
    closed-form kernels; slicing keeps the tag."""
    grouped = None

    def __getitem__(self, idx):
        out = list.__getitem__(self, idx)
        if isinstance(idx, slice):
            out = GraphMatrices(out)
            out.grouped = self.grouped
        return out


# ----------------------------------------------------------------------------
# lattice reductions on the GPU
# ----------------------------------------------------------------------------

def _lens_on(act_lens, device):
    return _native.lens_on(act_lens, device)


def _assert_sorted(act_lens):
    # fst_utils.py:382,432 — utterances are sorted by length, descending
    l = act_lens.tolist() if isinstance(act_lens, torch.Tensor) else list(act_lens)
    assert l == sorted(l, reverse=True)


_graph_cache = {}


def _device_graph(graph_matrices, device):
    """Graph matrices arrive as CPU int64 tensors every step
    (advanced_decoder.py:457-459); shared graphs (denominator) are the same
    tensor objects each time and are converted once."""
    if isinstance(graph_matrices, _native.Graph):
        return graph_matrices
    key = tuple((m.data_ptr(), m._version, tuple(m.shape)) for m in graph_matrices) + (str(device),)
    hit = _graph_cache.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[0], graph_matrices)):
        return hit[1]
    g = _native.Graph(graph_matrices, device)
    if graph_matrices[0].size(0) == 1:          # only shared graphs are worth caching
        if len(_graph_cache) > 16:
            _graph_cache.clear()
        _graph_cache[key] = (list(graph_matrices), g)
    return g


_grouped_cache = {}


def _device_grouped(graph_matrices, device):
    """GroupedGraph for matrices tagged by CTCGraphGen.get_decoding_matrices,
    else None."""
    st = getattr(graph_matrices, 'grouped', None)
    if st is None:
        return None
    key = (id(st), str(device))
    hit = _grouped_cache.get(key)
    if hit is None or hit[0] is not st:
        hit = (st, _native.GroupedGraph(st, device))
        _grouped_cache[key] = hit
    return hit[1]


class PathLogSumExp(torch.autograd.Function):
    """Forward-backward in the log semiring; same contract as the reference's
    PathLogSumExp (fst_utils.py:400-488): returns +logZ per utterance, the
    gradient is computed inside forward and scaled in backward.

    `negate` (an extension): return -logZ, the quantity a loss uses
    (advanced_decoder.py:486-496), with the occupancies stored negated by the kernel.  The
    backward pass then scales IN PLACE by the incoming gradient, which in a training step is
    exactly 1 for every utterance, and `asr_scale_rows_f32` skips those: the reference's
    `grad_output[None, :, None] * grads` (fst_utils.py:482-485) was a pass over the whole
    [T,B,C] tensor — 1.6 GB, 0.5 ms, for a bi-character alphabet at 512 utterances."""

    @staticmethod
    def forward(ctx, log_probs, act_lens, graph_matrices, neg_inf=-np.inf, negate=False):
        log_probs = log_probs.detach()
        sign = -1.0 if negate else 1.0
        _assert_sorted(act_lens)
        if not np.isfinite(neg_inf):
            neg_inf = NEG_INF
        lens = _lens_on(act_lens, log_probs.device)
        grouped = _device_grouped(graph_matrices, log_probs.device)
        if grouped is not None:                   # closed-form decoding graph
            log_cost, grads, _ = _native.grouped_fwbw(log_probs, lens, grouped, neg_inf)
            if negate:
                grads.neg_()
            ctx.grads = grads
            return -log_cost if negate else log_cost
        graph = _device_graph(graph_matrices, log_probs.device)
        if graph.dst_out is None:
            raise AssertionError("PathLogSumExp needs the 8 graph matrices")
        log_cost, grads, _ = _native.lattice_fwbw(log_probs, lens, graph, neg_inf, grad_sign=sign)
        ctx.grads = grads
        return -log_cost if negate else log_cost

    @staticmethod
    def backward(ctx, grad_output):
        grads, ctx.grads = ctx.grads, None
        if grads is None:
            raise RuntimeError("PathLogSumExp: backward a second time (the occupancies are scaled "
                               "in place and released after the first)")
        if grads.is_cuda and grads.dtype == torch.float32 and grads.dim() == 3:
            return (_native.scale_rows_(grads, grad_output), None, None, None, None)
        return (grad_output[None, :, None] * grads, None, None, None, None)


path_logsumexp = PathLogSumExp.apply


class NumeratorMinusDenominator(torch.autograd.Function):
    """-logZ(numerator lattices) + logZ(shared decoding graph) per utterance — the globally
    normalised loss of FSTDecoder.get_fst_loss (advanced_decoder.py:486-500:
    `num - den` with num = -path_reduction(numerator), den = -path_reduction(denominator)) —
    as ONE autograd node (an extension; two PathLogSumExp nodes give the same values): the
    numerator kernel writes its negated occupancies, the denominator kernel adds its own onto
    them in the same buffer (asr_lattice_grouped_fwbw_acc_f32), and the backward pass scales
    that buffer in place (factor 1 in a training step: no pass at all).  Two nodes meant two
    [T,B,C] tensors and autograd's addition of them: 7 GB of traffic per step for a bi-character
    alphabet at 768 utterances."""

    @staticmethod
    def forward(ctx, log_probs, act_lens, numerator, grouped, neg_inf):
        log_probs = log_probs.detach()
        _assert_sorted(act_lens)
        lens = _lens_on(act_lens, log_probs.device)
        graph = _device_graph(numerator, log_probs.device)
        # (the same constants as the two separate reductions: path_reduction evaluates training
        # lattices with NEG_INF and the decoding graph with the caller's neg_inf)
        zn, grads, _ = _native.lattice_fwbw(log_probs, lens, graph, NEG_INF, grad_sign=-1.0)
        zd, grads, _ = _native.grouped_fwbw(log_probs, lens, grouped, neg_inf, add_to=grads)
        ctx.grads = grads
        ctx.mark_non_differentiable(zn, zd)
        return zd - zn, zn, zd

    @staticmethod
    def backward(ctx, grad_output, _gn, _gd):
        grads, ctx.grads = ctx.grads, None
        if grads is None:
            raise RuntimeError("NumeratorMinusDenominator: backward a second time (the occupancies "
                               "are scaled in place and released after the first)")
        return (_native.scale_rows_(grads, grad_output), None, None, None, None)


class _PathViterbi(torch.autograd.Function):
    """max-plus alpha scan; d score / d log_probs is one-hot at the best path's
    input label of every active frame (what autograd gives the reference,
    fst_utils.py:366-370 + advanced_decoder.py:546-554)."""

    @staticmethod
    def forward(ctx, log_probs, act_lens, graph, neg_inf):
        lens = _lens_on(act_lens, log_probs.device)
        if isinstance(graph, _native.GroupedGraph):
            score, best = _native.grouped_forward(log_probs.detach(), lens, graph,
                                                  neg_inf, viterbi=True, want_path=True)
        else:
            score, best = _native.lattice_forward(log_probs.detach(), lens, graph,
                                                  neg_inf, viterbi=True, want_path=True)
        ctx.save_for_backward(best, lens)
        ctx.shape = log_probs.shape
        ctx.mark_non_differentiable(best)
        return score, best

    @staticmethod
    def backward(ctx, grad_score, _grad_best):
        best, lens = ctx.saved_tensors
        T, B, C = ctx.shape
        grad = torch.zeros(ctx.shape, dtype=grad_score.dtype, device=grad_score.device)
        mask = (torch.arange(T, device=lens.device)[:, None] < lens[None, :])
        vals = (grad_score[None, :] * mask).unsqueeze(-1)
        grad.scatter_(2, best.long().unsqueeze(-1), vals)
        return grad, None, None, None


def viterbi_path(log_probs, act_lens, graph_matrices, neg_inf=NEG_INF):
    """(score [B], best input label per frame [T,B] int32) of the best path."""
    graph = _device_grouped(graph_matrices, log_probs.device)
    if graph is None:
        graph = _device_graph(graph_matrices, log_probs.device)
    return _PathViterbi.apply(log_probs, act_lens, graph, neg_inf)


def path_reduction(log_probs, act_lens, graph_matrices, red_kind='logsumexp',
                   neg_inf=NEG_INF, negate=False):
    """Sum (logsumexp) or max over all paths through per-utterance graphs.
    Same dispatch as the reference (fst_utils.py:322-397).  `negate` (an extension): the
    negated result, cheaper than negating outside for the logsumexp reductions that carry
    their own gradient (see PathLogSumExp)."""
    if negate:
        if red_kind in ('logsumexp_fwb', 'logsumexp', 'logsumexp_autodiff') and log_probs.is_cuda and \
                (isinstance(graph_matrices, _native.Graph) or red_kind == 'logsumexp_fwb' or
                 len(graph_matrices) == 8):
            _assert_sorted(act_lens)
            return path_logsumexp(log_probs, act_lens, graph_matrices, NEG_INF, True)
        return -path_reduction(log_probs, act_lens, graph_matrices, red_kind, neg_inf)
    if isinstance(graph_matrices, _native.Graph):       # device-built lattice
        if red_kind in ('logsumexp_fwb', 'logsumexp', 'logsumexp_autodiff'):
            return path_logsumexp(log_probs, act_lens, graph_matrices, NEG_INF)
        assert red_kind in ['viterbi', 'viterbi_autodiff']
        _assert_sorted(act_lens)
        return _PathViterbi.apply(log_probs, act_lens, graph_matrices, neg_inf)[0]
    if (red_kind == 'logsumexp_fwb' or
            (red_kind == 'logsumexp' and len(graph_matrices) == 8)):
        return path_logsumexp(log_probs, act_lens, graph_matrices, NEG_INF)

    _, bs, _ = log_probs.size()
    assert graph_matrices[0].size(0) in [1, bs]
    assert all(sm.size(0) == graph_matrices[0].size(0) for sm in graph_matrices)
    _assert_sorted(act_lens)

    if red_kind in ['logsumexp', 'logsumexp_autodiff']:
        if not log_probs.requires_grad:
            lens = _lens_on(act_lens, log_probs.device)
            grouped = _device_grouped(graph_matrices, log_probs.device)
            if grouped is not None:
                return _native.grouped_forward(log_probs, lens, grouped, neg_inf)[0]
            graph = _device_graph(graph_matrices[:4], log_probs.device)
            return _native.lattice_forward(log_probs, lens, graph, neg_inf)[0]
        if getattr(graph_matrices, 'grouped', None) is not None:
            return path_logsumexp(log_probs, act_lens, graph_matrices, neg_inf)
        # differentiable: explicit beta scan needs the out-edge form
        if len(graph_matrices) == 4:
            graph_matrices = list(graph_matrices) + list(
                _cached_out_edges(graph_matrices, neg_inf))
        return path_logsumexp(log_probs, act_lens, graph_matrices, neg_inf)
    assert red_kind in ['viterbi', 'viterbi_autodiff']
    return viterbi_path(log_probs, act_lens, graph_matrices[:4], neg_inf)[0]


_out_edge_cache = {}


def _cached_out_edges(graph_matrices, nc_weight):
    key = tuple((m.data_ptr(), m._version, tuple(m.shape)) for m in graph_matrices[:4])
    hit = _out_edge_cache.get(key)
    if hit is None or not all(a is b for a, b in zip(hit[0], graph_matrices[:4])):
        if len(_out_edge_cache) > 16:
            _out_edge_cache.clear()
        hit = (list(graph_matrices[:4]),
               in_to_out_edge_matrices(graph_matrices, nc_weight))
        _out_edge_cache[key] = hit
    return hit[1]


# ----------------------------------------------------------------------------
# batching (reference fst_utils.py:491-521)
# ----------------------------------------------------------------------------

def batch_training_graph_matrices(matrices, nc_weight=NEG_INF, device='cpu'):
    bs = len(matrices)
    max_n = max([m[0].size(0) for m in matrices])
    max_ks = [max([m[i].size(1) for m in matrices])
              for i in range(len(matrices[0]))]
    batched_matrices = []
    for i, m in enumerate(matrices[0]):
        batched_matrices.append(torch.full(
            (bs, max_n, max_ks[i]),
            0 if m.dtype == torch.int64 else nc_weight,
            dtype=m.dtype, device=device))
    for b, ms in enumerate(matrices):
        for i, m in enumerate(ms):
            batched_matrices[i][b, :m.size(0), :m.size(1)] = m
    return batched_matrices


def make_full_ngram_table(context_order, num_symbols, num_classes):
    """reference fst_utils.py:524-534"""
    if num_symbols is None:
        num_symbols = int(round(num_classes ** (1.0 / context_order)))
    num_classes = num_symbols ** context_order
    idx = np.arange(num_classes)[:, None]
    pw = num_symbols ** np.arange(context_order)[::-1][None, :]
    ngram_to_class = torch.from_numpy((idx // pw % num_symbols).astype(np.int64))
    return num_symbols, num_classes, ngram_to_class


def setattr_matched(obj, var, val):
    val_orig = getattr(obj, var)
    if val_orig is not None:
        assert val_orig == val, ("The value of %s is %s but should be %s" %
                                 (var, val_orig, val))
    else:
        setattr(obj, var, val)


# ----------------------------------------------------------------------------
# closed-form CTC graphs
# ----------------------------------------------------------------------------

class DecodingTransducer(object):
    """The CTC decoding FST (reference build_ctc_mono_decoding_fst
    fst_utils.py:679-726 / build_ctc_bigram_decoding_fst :729-835) as a
    transition function instead of an OpenFst object: input-deterministic, all
    states final, start state 0.  `arcs()` enumerates (src, dst, ilabel0,
    olabel) with 0-based input labels (arc.ilabel - 1, fst_utils.py:270)."""

    def __init__(self, num_symbols, context_order, allow_nonblank_selfloops=True,
                 use_contextual_blanks=False,
                 loop_using_symbol_repetitions=False,
                 eval_repeats_in_context=False):
        assert not (eval_repeats_in_context and loop_using_symbol_repetitions)
        assert not eval_repeats_in_context                     # fst_utils.py:760
        if loop_using_symbol_repetitions:
            raise NotImplementedError(
                "loop_using_symbol_repetitions graphs are not built by the "
                "closed-form generator (no shipped YAML uses them)")
        self.S = num_symbols
        self.context_order = context_order
        self.allow_nonblank_selfloops = allow_nonblank_selfloops
        self.use_contextual_blanks = use_contextual_blanks
        self.num_states = num_symbols ** context_order

    def arcs(self):
        S = self.S
        if self.context_order == 1:
            src = np.repeat(np.arange(S), S)
            dst = np.tile(np.arange(S), S)
            il = dst.copy()
            ol = np.where(src == dst, 0, dst)
            return src, dst, il, ol
        s1 = np.arange(S * S)
        c1, l1 = s1 // S, s1 % S
        c2 = np.where(l1 == 0, c1, l1)                          # (:798-801)
        src = np.repeat(s1, S)
        l2 = np.tile(np.arange(S), S * S)
        c2r = np.repeat(c2, S)
        dst = c2r * S + l2
        il = np.where((l2 != 0) | self.use_contextual_blanks, c2r * S + l2, 0)
        ol = np.where((l2 == 0) | (src == dst), 0, l2)
        if self.allow_nonblank_selfloops:                       # (:791-795)
            sl = s1[(l1 != 0) & (c1 != l1)]
            src = np.concatenate([src, sl])
            dst = np.concatenate([dst, sl])
            il = np.concatenate([il, sl])
            ol = np.concatenate([ol, np.zeros_like(sl)])
        return src, dst, il, ol

    def grouped_structure(self, nc_weight=NEG_INF):
        """Group factorisation of the graph (see include/asr_amd.h,
        asr_lattice_grouped_*): state s1 feeds group g_of[s1]; state s2 accepts
        from every s1 with g_of[s1] == h_of[s2] plus its extra self-loop."""
        S, N = self.S, self.num_states
        s = np.arange(N)
        if self.context_order == 1:
            g_of = np.zeros(N, np.int64)
            h_of = np.zeros(N, np.int64)
            label = s.copy()
            selfx = np.zeros(N, np.int64)
            G = 1
        else:
            c, l = s // S, s % S
            g_of = np.where(l == 0, c, l)
            h_of = c
            label = np.where((l != 0) | self.use_contextual_blanks, c * S + l, 0)
            selfx = ((l != 0) & (c != l)).astype(np.int64) * int(self.allow_nonblank_selfloops)
            G = S

        def members(of):
            lists = [np.nonzero(of == grp)[0] for grp in range(G)]
            w = max(1, max(len(x) for x in lists))
            m = np.full((G, w), -1, np.int64)
            for grp, x in enumerate(lists):
                m[grp, :len(x)] = x
            return m
        counts = np.bincount(label, minlength=S ** self.context_order)
        return dict(g_of=g_of, h_of=h_of, label=label, selfx=selfx,
                    uniq=(counts[label] == 1).astype(np.int64),
                    mem_g=members(g_of), mem_h=members(h_of),
                    term=np.zeros(N, np.float32))

    def transition_tables(self):
        """dense [num_states, num_classes] next-state / output tables
        (-1 = no arc) for walking label sequences."""
        if not hasattr(self, '_tables'):
            C = self.S ** self.context_order
            src, dst, il, ol = self.arcs()
            nxt = np.full((self.num_states, C), -1, np.int64)
            out = np.zeros((self.num_states, C), np.int64)
            nxt[src, il] = dst
            out[src, il] = ol
            self._tables = (nxt, out)
        return self._tables

    def read_out(self, ilabels):
        """Output labels produced while consuming 0-based input labels from
        the start state — FSTDecoder.decode's compose + shortestpath read-out
        (advanced_decoder.py:556-571)."""
        nxt, out = self.transition_tables()
        s, res = 0, []
        for il in ilabels:
            il = int(il)
            ns = nxt[s, il]
            if ns < 0:
                return []
            if out[s, il] > 0:
                res.append(int(out[s, il]))
            s = ns
        return res


def ctc_training_arcs(labels, label_lens, num_symbols, context_order,
                      allow_nonblank_selfloops=True, use_contextual_blanks=False):
    """Closed form of compose(decoding_fst, chain(labels)) for a whole batch.

    labels [B,Lmax] (symbols, already reduced modulo num_symbols), label_lens
    [B].  States of utterance b: 0 = initial blank, 2j+1 = label j, 2j+2 = the
    blank after it; N_b = 2 L_b + 1.  Returns flat arrays
    (b, src, dst, ilabel) and the per-utterance state counts."""
    S = num_symbols
    labels = np.asarray(labels, np.int64)
    lens = np.asarray(label_lens, np.int64)
    B, Lmax = labels.shape if labels.ndim == 2 else (len(lens), 0)
    j = np.arange(Lmax)[None, :]
    valid = j < lens[:, None]                                  # label j exists
    l = np.where(valid, labels, 0)
    ctx = np.concatenate([np.zeros((B, 1), np.int64), l[:, :-1]], 1) if Lmax else l
    if context_order == 1:
        emit = l
        skip_ok = l[:, :-1] != l[:, 1:] if Lmax > 1 else np.zeros((B, 0), bool)
        selfloop = np.ones_like(valid)
        blank_of_label = np.zeros_like(l)       # blank after label j
        blank0 = np.zeros(B, np.int64)
    else:
        emit = ctx * S + l
        if Lmax > 1:
            skip_ok = ~((ctx[:, :-1] == l[:, :-1]) & (l[:, :-1] == l[:, 1:]))
        else:
            skip_ok = np.zeros((B, 0), bool)
        selfloop = np.full(valid.shape, bool(allow_nonblank_selfloops)) | (ctx == l)
        blank_of_label = l * S if use_contextual_blanks else np.zeros_like(l)
        blank0 = np.zeros(B, np.int64)
    bb = np.broadcast_to(np.arange(B)[:, None], valid.shape)
    jj = np.broadcast_to(j, valid.shape)
    parts = []

    def add(mask, src, dst, il):
        parts.append((bb[mask], src[mask], dst[mask], il[mask]))
    # blank -> next label   (2j -> 2j+1)
    add(valid, 2 * jj, 2 * jj + 1, emit)
    # label self-loop
    add(valid & selfloop, 2 * jj + 1, 2 * jj + 1, emit)
    # label -> following blank
    add(valid, 2 * jj + 1, 2 * jj + 2, blank_of_label)
    # following blank self-loop
    add(valid, 2 * jj + 2, 2 * jj + 2, blank_of_label)
    # label -> next label
    if Lmax > 1:
        m = valid[:, 1:] & skip_ok
        parts.append((bb[:, :-1][m], (2 * jj[:, :-1] + 1)[m], (2 * jj[:, :-1] + 3)[m],
                      emit[:, 1:][m]))
    # initial blank self-loop
    parts.append((np.arange(B), np.zeros(B, np.int64), np.zeros(B, np.int64), blank0))
    b_, s_, d_, i_ = [np.concatenate(x) for x in zip(*parts)]
    return b_, s_, d_, i_, 2 * lens + 1


class BaseGraphGen(object):
    """reference fst_utils.py:546-676 (grammar-FST composition excluded: it
    needs an external LM FST)."""

    def __init__(self, num_symbols=None, num_classes=None,
                 ngram_to_class_file=None, context_order=None,
                 nc_weight=NEG_INF, for_forward_only=False,
                 grammar_fst=None, vocabulary=None, **kwargs):
        super(BaseGraphGen, self).__init__(**kwargs)
        if ngram_to_class_file is not None or grammar_fst is not None:
            raise NotImplementedError(
                "ngram_to_class_file / grammar_fst graphs need OpenFst")
        self.num_classes = num_classes
        self.num_symbols = num_symbols
        self.context_order = context_order
        if self.context_order is None:
            self.context_order = 1
        (num_symbols, num_classes, ngram_to_class) = make_full_ngram_table(
            self.context_order, num_symbols, num_classes)
        setattr_matched(self, 'num_symbols', num_symbols)
        setattr_matched(self, 'num_classes', num_classes)
        self.ngram_to_class = ngram_to_class
        self.ngrams = ngram_to_class.tolist()
        self.grammar_fst_path = None
        self.nc_weight = nc_weight
        self.decoding_fst = self.get_decoding_fst()
        self.decoding_mats = {}
        self.for_forward_only = for_forward_only

    def get_decoding_fst(self):
        return self.get_hc_fst()

    def get_hc_fst(self):
        raise NotImplementedError()

    def _reduce_labels(self, labels):                           # (:592-600)
        if isinstance(labels, torch.Tensor):
            labels = labels.cpu().numpy()
        labels = np.asarray(labels)
        if np.any(labels > self.num_symbols):
            labels = labels % self.num_symbols
            if getattr(self, '_print_bigram_data_warn', True):
                print("Got input symbol larger than num_symbols. "
                      "Are you using a Bigram dataset with FSTs?")
                self._print_bigram_data_warn = False
        return labels

    def get_training_matrices_batch(self, labels, label_lens, device='cpu'):
        """(:607-613) — all utterances at once; returns 4 or 8 tensors
        [B, maxN, K] padded like batch_training_graph_matrices."""
        if isinstance(label_lens, torch.Tensor):
            label_lens = label_lens.cpu().numpy()
        label_lens = np.asarray(label_lens, np.int64)
        B = len(label_lens)
        if isinstance(labels, (list, tuple)):
            lmax = int(label_lens.max()) if B else 0
            lab = np.zeros((B, lmax), np.int64)
            for i, row in enumerate(labels):
                row = np.asarray(row.cpu() if isinstance(row, torch.Tensor) else row)
                lab[i, :label_lens[i]] = row[:label_lens[i]]
            labels = lab
        labels = self._reduce_labels(labels)
        labels = labels[:, :int(label_lens.max()) if B else 0]
        d = self.decoding_fst
        b_, s_, d_, i_, n_states = ctc_training_arcs(
            labels, label_lens, self.num_symbols, self.context_order,
            d.allow_nonblank_selfloops, d.use_contextual_blanks)
        nmax = int(n_states.max())
        w_ = np.zeros(len(b_), np.float32)
        term = np.full((B, nmax, 1), self.nc_weight, np.float32)
        bi = np.arange(B)
        term[bi, n_states - 1, 0] = 0.0            # blank after the last label
        has = label_lens > 0
        term[bi[has], n_states[has] - 2, 0] = 0.0  # last label state
        mats = self._batched(B, nmax, d_, s_, i_, w_, b_) + [torch.from_numpy(term)]
        if not self.for_forward_only:
            mats += self._batched(B, nmax, s_, d_, i_, w_, b_) + [
                torch.from_numpy(term.copy())]
        if str(device) != 'cpu':
            mats = [m.to(device) for m in mats]
        return mats

    def get_training_graph_device(self, labels, label_lens, device):
        """Same lattices as get_training_matrices_batch, built ON the device
        (SURVEY.md §8f N2): only the labels cross PCIe.  Returns the
        device-resident graph object path_reduction accepts in place of the 8
        tensors."""
        if isinstance(label_lens, torch.Tensor):
            label_lens = label_lens.cpu().numpy()
        label_lens = np.asarray(label_lens, np.int64)
        B = len(label_lens)
        lmax = int(label_lens.max()) if B else 0
        if isinstance(labels, (list, tuple)):
            lab = np.zeros((B, lmax), np.int64)
            for i, row in enumerate(labels):
                row = np.asarray(row.cpu() if isinstance(row, torch.Tensor) else row)
                lab[i, :label_lens[i]] = row[:label_lens[i]]
            labels = lab
        labels = self._reduce_labels(labels)[:, :lmax]
        d = self.decoding_fst
        lab_d = torch.as_tensor(np.ascontiguousarray(labels)).to(device, torch.int32)
        len_d = torch.as_tensor(label_lens).to(device, torch.int32)
        return _native.build_ctc_graph(lab_d, len_d, self.num_symbols, self.context_order,
                                       d.allow_nonblank_selfloops, d.use_contextual_blanks,
                                       self.nc_weight)

    def _batched(self, B, nmax, own, other, il, w, b):
        st, ilab, wt = _arcs_to_matrices(B * nmax, b * nmax + own, other, il, w,
                                         self.nc_weight)
        k = st.shape[1]
        return [torch.from_numpy(st.reshape(B, nmax, k)),
                torch.from_numpy(ilab.reshape(B, nmax, k)),
                torch.from_numpy(wt.reshape(B, nmax, k))]

    def get_training_matrices(self, labels, device='cpu'):      # (:647-660)
        labels = np.asarray(labels.cpu() if isinstance(labels, torch.Tensor)
                            else labels).reshape(1, -1)
        mats = self.get_training_matrices_batch(labels, [labels.shape[1]], device)
        return tuple(m[0] for m in mats)

    def get_decoding_matrices(self, device='cpu', out_edges=False):  # (:662-676)
        key = str(device)
        ret = self.decoding_mats.get(key)
        if ret is not None:
            return ret
        d = self.decoding_fst
        src, dst, il, _ = d.arcs()
        mats = arcs_to_graph_matrices(
            d.num_states, src, dst, il, np.zeros(len(src), np.float32),
            np.zeros(d.num_states, np.float32), self.nc_weight,
            self.for_forward_only)
        tagged = GraphMatrices(m.unsqueeze(0).to(device) for m in mats)
        tagged.grouped = d.grouped_structure(self.nc_weight)
        self.decoding_mats[key] = tagged
        return self.decoding_mats[key]


class CTCGraphGen(BaseGraphGen):
    """reference fst_utils.py:1053-1068 (context orders 1 and 2)."""

    def __init__(self, context_order=None, graph_build_args=None, **kwargs):
        assert context_order in (1, 2, 3)
        if context_order == 3:
            raise NotImplementedError("trigram CTC graphs are out of scope")
        self.graph_build_args = graph_build_args or {}
        super(CTCGraphGen, self).__init__(context_order=context_order, **kwargs)

    def get_hc_fst(self):
        args = self.graph_build_args if self.context_order == 2 else {}
        return DecodingTransducer(self.num_symbols, self.context_order, **args)


# ----------------------------------------------------------------------------
# Hypothesis state sets on a language-model FST (reference fst_utils.py:23-188:
# score_nodes, reduce_weights, expand_epsilon, expand_non_epsilon, expand,
# expand_all).  A hypothesis of the LM-fused beam search is a bag
# {LM state: cost} (cost = -log p); one decoder step pushes every bag through
# the arcs of one input label and through the epsilon (back-off) arcs behind
# them.  The reference walks Python dicts arc by arc, one beam and one label at
# a time; here the LM is a CSR array bundle (att_speech/lm_fst.py) and ALL beams
# and ALL labels of a step are expanded together: arcs gathered with one index
# expression, equal (beam, label, state) targets merged with a sort + segmented
# logaddexp / min, the epsilon closure done level by level in the topological
# rank of the LM's epsilon graph (so every state's cost is final when it is
# pushed on).  The dict functions of the reference keep their names and
# semantics on top of that core.
# ----------------------------------------------------------------------------
def reduce_weights(ws, use_log_probs):
    """-log(sum exp(-w)) if use_log_probs else min(w); inf for an empty list (:46-58)."""
    ws = np.asarray(list(ws), np.float64)
    if ws.size == 0:
        return float('inf')
    if use_log_probs:
        return float(-np.logaddexp.reduce(-ws))
    return float(ws.min())


def _reduce_by_key(key, cost, use_log_probs):
    if key.size == 0:
        return key, cost
    order = np.argsort(key, kind='stable')
    k, c = key[order], cost[order]
    idx = np.nonzero(np.r_[True, k[1:] != k[:-1]])[0]
    if use_log_probs:
        with np.errstate(invalid='ignore'):
            red = -np.logaddexp.reduceat(-c, idx)
    else:
        red = np.minimum.reduceat(c, idx)
    return k[idx], red


def _gather_arcs(lo, hi):
    cnt = hi - lo
    total = int(cnt.sum())
    owner = np.repeat(np.arange(len(lo)), cnt)
    offs = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    return owner, np.repeat(lo, cnt) + offs


def _expand_epsilon_dfs(LG, nodes, use_log_probs):
    """Reference algorithm (:86-137) for LMs whose epsilon graph has a cycle
    somewhere: depth-first topological order over the part reachable from
    `nodes`; a cycle reachable from them is an error."""
    preds, done, order = {}, set(), []
    for s0 in nodes:
        preds.setdefault(s0, []).append((-1, nodes[s0]))
        if s0 in done:
            continue
        stack = [(s0, iter(range(int(LG.ptr[s0]), int(LG.ptr_ne[s0]))))]
        on_stack = {s0}
        while stack:
            s, it = stack[-1]
            for a in it:
                n = int(LG.dst[a])
                preds.setdefault(n, []).append((s, float(LG.weight[a])))
                if n in done:
                    continue
                if n in on_stack:
                    raise ValueError("LG has epsilon-cycles!")
                on_stack.add(n)
                stack.append((n, iter(range(int(LG.ptr[n]), int(LG.ptr_ne[n])))))
                break
            else:
                stack.pop()
                on_stack.discard(s)
                done.add(s)
                order.append(s)
    out = {-1: 0.0}
    for s in order[::-1]:
        out[s] = reduce_weights([out[p] + w for p, w in preds[s]], use_log_probs)
    del out[-1]
    return out


def _epsilon_closure(LG, grp, st, w, use_log_probs):
    """grp/st/w: parallel arrays, (grp, st) unique.  Returns them closed under the
    epsilon arcs, sorted by (grp, st)."""
    S = LG.num_states()
    key, w = _reduce_by_key(grp * S + st, w, use_log_probs)
    if key.size == 0:
        return grp[:0], st[:0], w
    rank = LG.eps_rank()
    if rank is None:
        g_all, s_all, w_all = [], [], []
        grp, st = key // S, key % S
        for g in np.unique(grp):
            m = grp == g
            d = _expand_epsilon_dfs(LG, dict(zip(st[m].tolist(), w[m].tolist())),
                                    use_log_probs)
            g_all.append(np.full(len(d), g, np.int64))
            s_all.append(np.fromiter(d.keys(), np.int64, len(d)))
            w_all.append(np.fromiter(d.values(), np.float64, len(d)))
        key, w = _reduce_by_key(np.concatenate(g_all) * S + np.concatenate(s_all),
                                np.concatenate(w_all), use_log_probs)
        return key // S, key % S, w
    level, top = 0, int(rank.max())
    while level <= top:
        st = key % S
        sel = np.nonzero((rank[st] == level) & (LG.ptr_ne[st] > LG.ptr[:-1][st]))[0]
        if sel.size:
            owner, arcs = _gather_arcs(LG.ptr[:-1][st[sel]], LG.ptr_ne[st[sel]])
            nk = (key[sel] // S)[owner] * S + LG.dst[arcs]
            key, w = _reduce_by_key(np.concatenate([key, nk]),
                                    np.concatenate([w, w[sel][owner] + LG.weight[arcs]]),
                                    use_log_probs)
        level += 1
    return key // S, key % S, w


def expand_all_batched(LG, num_classes, grp, st, w, use_log_probs=False):
    """Push the bags `grp` (any non-negative ids, e.g. beam indices) through the arcs
    of EVERY input label at once.  Returns (bag, state, cost) arrays sorted by bag
    with bag = grp * num_classes + ilabel.  Like the reference's expand_all
    (:163-185) the arcs of label 0 form bag `grp * num_classes + 0`, and a label
    >= num_classes is an IndexError."""
    grp = np.asarray(grp, np.int64); st = np.asarray(st, np.int64)
    w = np.asarray(w, np.float64)
    owner, arcs = _gather_arcs(LG.ptr[:-1][st], LG.ptr[1:][st])
    il = LG.ilabel[arcs]
    if il.size and int(il.max()) >= num_classes:
        raise IndexError("LM input label %d outside the %d classes" % (il.max(), num_classes))
    bag = grp[owner] * num_classes + il
    return _epsilon_closure(LG, bag, LG.dst[arcs], w[owner] + LG.weight[arcs], use_log_probs)


def _bag_arrays(nodes):
    n = len(nodes)
    return (np.fromiter(nodes.keys(), np.int64, n), np.fromiter(nodes.values(), np.float64, n))


def expand_epsilon(LG, nodes, use_log_probs):
    """All states reachable from the bag over epsilon arcs, costs summed over the
    epsilon paths (:86-137)."""
    st, w = _bag_arrays(nodes)
    _, st, w = _epsilon_closure(LG, np.zeros(len(st), np.int64), st, w, use_log_probs)
    return dict(zip(st.tolist(), w.tolist()))


def expand_non_epsilon(LG, nodes, label, use_log_probs=False):
    """(:140-154)"""
    st, w = _bag_arrays(nodes)
    owner, arcs = _gather_arcs(LG.ptr[:-1][st], LG.ptr[1:][st])
    m = LG.ilabel[arcs] == label
    k, c = _reduce_by_key(LG.dst[arcs][m], (w[owner] + LG.weight[arcs])[m], use_log_probs)
    return dict(zip(k.tolist(), c.tolist()))


def expand(LG, nodes, label, use_log_probs=False):
    """(:157-161)"""
    return expand_epsilon(LG, expand_non_epsilon(LG, nodes, label, use_log_probs),
                          use_log_probs)


def expand_all(LG, num_classes, nodes, use_log_probs=False):
    """One bag per input label (:163-185)."""
    st, w = _bag_arrays(nodes)
    bag, st, w = expand_all_batched(LG, num_classes, np.zeros(len(st), np.int64), st, w,
                                    use_log_probs)
    out = [{} for _ in range(num_classes)]
    for b, s, c in zip(bag.tolist(), st.tolist(), w.tolist()):
        out[b][s] = c
    return out


def score_nodes(LG, nodes, final=False, use_log_probs=False, expand_symbol=None):
    """Cost of all paths ending in the bag; final: optionally extend by
    `expand_symbol`, then count paths into final states only (:23-43)."""
    if final:
        if expand_symbol is not None:
            after = expand(LG, nodes, LG.input_symbols().find(expand_symbol))
            assert not set(after).intersection(set(nodes))
            after.update(nodes)
            nodes = after
        ws = [LG.final(k) + v for k, v in nodes.items()]
    else:
        ws = list(nodes.values())
    return reduce_weights(ws, use_log_probs)
