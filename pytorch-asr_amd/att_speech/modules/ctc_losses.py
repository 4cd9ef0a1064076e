"""att_speech.modules.ctc_losses — loss front-ends of the reference
(att_speech/modules/ctc_losses.py) on the MI355X kernels.

Kept: get_normalized_acts (:29-43), ctc_loss (:46-64, evaluated with the
lattice kernel instead of F.ctc_loss, same 'mean' reduction), ctc_fst_loss
(:617-657), and ctc_raw_loss / ctc_raw_loss_batch (:521-609) as front-ends of the
same sparse lattice kernel: the reference evaluates those two with dense
[N, N] transition matrices (RawGenericCTC*, :268-390, O(T N^2)); the value is the
same sum over alignments, so no dense path is built here — the dense
arithmetic lives in the golden fixtures as an independent oracle."""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch

from att_speech import _native
from att_speech.fst_utils import CTCGraphGen, path_reduction, path_logsumexp  # noqa: F401


class _GroupLogSoftmax(torch.autograd.Function):
    """log_softmax over contiguous groups of `group` classes (HIP fwd + bwd)."""

    @staticmethod
    def forward(ctx, acts, group):
        y = _native.log_softmax_fwd(acts.contiguous(), group)
        ctx.save_for_backward(y)
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _native.log_softmax_bwd(y, dy.contiguous(), ctx.group), None


def get_normalized_acts(acts, acts_lens, num_symbols, context_order,
                        normalize_by_dim, normalize_logits=True):
    """reference ctc_losses.py:29-43.  normalize_by_dim > 0 normalises over
    axis normalize_by_dim + 2 of the [T,B,S,...,S] view.  The last axis (contiguous
    groups of S: what the shipped configs use) goes straight to the group kernel; any
    other axis (context_order >= 3) is moved to the end, normalised there and moved back."""
    assert context_order == 1 or num_symbols
    del acts_lens  # unused
    if normalize_by_dim:
        assert acts.size(-1) == num_symbols ** context_order
        if not 0 < normalize_by_dim < context_order:
            raise IndexError("normalize_by_dim=%d: the [T,B,S,...] view of context_order %d has "
                             "no axis %d" % (normalize_by_dim, context_order, normalize_by_dim + 2))
        if normalize_by_dim == context_order - 1:
            return _GroupLogSoftmax.apply(acts, num_symbols)
        size = acts.size()
        v = acts.view(*(tuple(size[:2]) + (num_symbols,) * context_order))
        v = v.movedim(normalize_by_dim + 2, -1).contiguous()
        y = _GroupLogSoftmax.apply(v.view(size[0], size[1], -1), num_symbols)
        return y.view(v.shape).movedim(-1, normalize_by_dim + 2).contiguous().view(size)
    elif normalize_logits:
        return _GroupLogSoftmax.apply(acts, acts.size(-1))
    return acts


def _labels_to_batch(labels, label_lens):
    """flat concatenated labels -> padded [B, Lmax] (reference :646-652)."""
    label_lens = torch.as_tensor(label_lens)
    lab = torch.zeros((label_lens.numel(), int(label_lens.max()) if label_lens.numel() else 0),
                      dtype=torch.int64)
    ls = 0
    for b in range(label_lens.numel()):
        le = ls + int(label_lens[b])
        lab[b, :le - ls] = torch.as_tensor(labels[ls:le]).long()
        ls = le
    return lab


_graph_gens = {}


def _graph_gen(num_symbols, context_order, **graph_build_args):
    key = (num_symbols, context_order, tuple(sorted(graph_build_args.items())))
    if key not in _graph_gens:
        _graph_gens[key] = CTCGraphGen(
            context_order=context_order, num_symbols=num_symbols,
            for_forward_only=False, graph_build_args=graph_build_args)
    return _graph_gens[key]


def ctc_fst_loss(acts, labels, act_lens, label_lens,
                 num_symbols=0, context_order=1, normalize_by_dim=None,
                 allow_nonblank_selfloops=True,
                 loop_using_symbol_repetitions=False,
                 other_data_in_batch=None,
                 eval_repeats_in_context=False,
                 neg_inf=-1e20):
    """reference ctc_losses.py:617-657: per-utterance -log p(labels | acts)."""
    log_probs = get_normalized_acts(acts, act_lens, num_symbols,
                                    context_order, normalize_by_dim,
                                    normalize_logits=True)
    assert not (eval_repeats_in_context and loop_using_symbol_repetitions)
    S = log_probs.size(2)
    if other_data_in_batch and 'graph_matrices' in other_data_in_batch:
        graph_matrices = other_data_in_batch['graph_matrices']
    else:
        gg_kwargs = {}
        if context_order == 2:
            S = int(round(np.sqrt(S)))
            gg_kwargs = dict(
                allow_nonblank_selfloops=allow_nonblank_selfloops,
                loop_using_symbol_repetitions=loop_using_symbol_repetitions,
                eval_repeats_in_context=eval_repeats_in_context)
        graph_gen = _graph_gen(S, context_order, **gg_kwargs)
        labels_b = _labels_to_batch(labels, label_lens)
        graph_matrices = graph_gen.get_training_matrices_batch(labels_b, label_lens)
    return path_reduction(log_probs, act_lens, graph_matrices, neg_inf=neg_inf, negate=True)


def ctc_loss(acts, labels, act_lens, label_lens,
             num_symbols=None, context_order=1, normalize_by_dim=None,
             allow_nonblank_selfloops=True,
             loop_using_symbol_repetitions=False,
             eval_repeats_in_context=False,
             other_data_in_batch=None):
    """reference ctc_losses.py:46-64 — F.ctc_loss(log_softmax(acts), ...,
    reduction='mean'): per-utterance losses divided by the target lengths,
    then averaged.  Evaluated with the mono CTC lattice (identical value,
    SURVEY.md §4) instead of the vendor CTC."""
    for condition in [context_order == 1,
                      normalize_by_dim is None,
                      not eval_repeats_in_context,
                      allow_nonblank_selfloops,
                      not loop_using_symbol_repetitions,
                      not other_data_in_batch]:
        assert condition, "Option not supported in this loss"
    assert acts.size(0) == act_lens[0]
    assert int(torch.as_tensor(labels).max()) < acts.size(2)
    losses = ctc_fst_loss(acts, labels, act_lens, label_lens,
                          num_symbols=acts.size(2), context_order=1,
                          normalize_by_dim=None)
    # an utterance with no feasible alignment: the lattice returns -neg_inf-sized values,
    # F.ctc_loss(zero_infinity=False) returns inf (and the mean with it)
    losses = torch.where(losses >= 5e19, torch.full_like(losses, float('inf')), losses)
    tl = torch.as_tensor(label_lens).to(losses.device, losses.dtype).clamp(min=1)
    return (losses / tl).mean()


def ctc_raw_loss_batch(acts, labels, act_lens, label_lens,
                       num_symbols=0, context_order=1, normalize_by_dim=None,
                       allow_nonblank_selfloops=True,
                       loop_using_symbol_repetitions=False,
                       eval_repeats_in_context=False,
                       other_data_in_batch=None,
                       **kwargs):
    """reference ctc_losses.py:563-609 — per-utterance CTC losses over the dense
    mono / bicontext transition matrices (get_CTC_matrices_mono :67-94,
    get_CTC_matrices_bicontext :111-166).  Same alignments as the sparse training
    lattice of CTCGraphGen, so it is evaluated with the lattice kernel.  The bicontext
    matrices with eval_repeats_in_context=False treat a repeated symbol differently
    from the FST lattice; that combination is refused for label
    sequences that contain a repeat instead of returning a different number."""
    assert not other_data_in_batch
    assert not (eval_repeats_in_context and loop_using_symbol_repetitions)
    if kwargs:
        raise NotImplementedError("ctc_raw_loss: unsupported options %s" % sorted(kwargs))
    if context_order == 1:
        assert not loop_using_symbol_repetitions
        return ctc_fst_loss(acts, labels, act_lens, label_lens, num_symbols=num_symbols,
                            context_order=1, normalize_by_dim=normalize_by_dim)
    assert context_order == 2 and normalize_by_dim == 1 and num_symbols > 0
    if loop_using_symbol_repetitions:
        raise NotImplementedError("loop_using_symbol_repetitions lattices are not built")
    if not eval_repeats_in_context:
        flat = torch.as_tensor(labels).long().cpu() % num_symbols
        ends = torch.as_tensor(label_lens).long().cpu().cumsum(0).tolist()
        start = 0
        for end in ends:
            seq = flat[start:end]
            if seq.numel() > 1 and bool((seq[1:] == seq[:-1]).any()):
                raise NotImplementedError(
                    "dense bicontext CTC with eval_repeats_in_context=False on a label "
                    "sequence with a repeated symbol differs from the FST lattice")
            start = end
    # the bicontext matrices emit a CONTEXT-SPECIFIC blank before each symbol
    # (get_CTC_matrices_bicontext, :123-166): the training lattice with contextual blanks
    log_probs = get_normalized_acts(acts, act_lens, num_symbols, 2, normalize_by_dim,
                                    normalize_logits=True)
    graph_gen = _graph_gen(num_symbols, 2, allow_nonblank_selfloops=allow_nonblank_selfloops,
                           use_contextual_blanks=True)
    graph_matrices = graph_gen.get_training_matrices_batch(
        _labels_to_batch(labels, label_lens), label_lens)
    return path_reduction(log_probs, act_lens, graph_matrices, negate=True)


# the reference's per-utterance Python loop (:521-560) computes the same values
ctc_raw_loss = ctc_raw_loss_batch
