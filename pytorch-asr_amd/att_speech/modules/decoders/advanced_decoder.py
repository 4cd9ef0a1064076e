"""att_speech.modules.decoders.advanced_decoder — the CTC / FST decoders of the
reference module of the same dotted name (att_speech/modules/decoders/
advanced_decoder.py) on the MI355X kernels.

What is kept is the surface a YAML, a checkpoint or `decode.py` can see
(SURVEY.md §8b): class names, constructor keywords, `forward / decode / logits`
signatures, the returned dicts, and the parameter names below `fc.0.module.0`.
How it is organised is this build's own:

* a class-projection layer is an object with `class_weight_bias()` — the
  `[C, F]` prototype matrix and `[C]` bias of THIS step (tying, weight noise and,
  for `NGramLinear`, the embedding network already applied) — and the frame
  product is `project_frames`, one autograd function for every embedder;
* `NGramLinear` keeps its computed prototype matrix across the passes of an
  evaluation (no autograd) while the parameters are unchanged;
* the two decoders share `_ProjectionDecoder` (construction of `fc`, the
  optional outputs of `decode`);
* normalisation, row-max stabilisation, lattice reductions, arg-max and Viterbi
  run in the HIP kernels behind include/asr_amd.h.
"""
from __future__ import absolute_import, division, print_function

import os

import torch
import torch.nn.functional as F
from torch import nn

from att_speech import _native, fst_utils, utils
from att_speech.logger import DefaultTensorLogger
from att_speech.modules.common import SequenceWise
from att_speech.modules.ctc_losses import (  # noqa: F401  (resolved by name from YAML)
    ctc_fst_loss, ctc_loss, ctc_raw_loss, ctc_raw_loss_batch, get_normalized_acts)
from att_speech.modules.decoders.base_decoder import BaseDecoder

logger = DefaultTensorLogger()


# ---------------------------------------------------------------------------
# frame product  y[r, c] = sum_f x[r, f] W[c, f] + b[c]
# ---------------------------------------------------------------------------
def _frame_chunks(rows):
    """Largest power-of-two chunk count (<= 256) that divides `rows` and leaves
    at least 64 frames per chunk; 1 if there is none."""
    for g in (256, 128, 64, 32, 16, 8, 4, 2):
        if rows % g == 0 and rows // g >= 64:
            return g
    return 1


class _FrameProjection(torch.autograd.Function):
    """The class projection of a whole batch of frames (very many rows, few or
    many classes).  Forward is one dense product.  Backward needs two reductions
    over ALL frames into a small output; as single library calls they occupy 49
    workgroups (dW, 0.5 ms at 171 k frames) and ONE workgroup (db, 0.86 ms), so
    both are split over chunks of frames: a batched product and a two-stage sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.with_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        rows = dy.numel() // dy.size(-1)
        dy_r, x_r = dy.reshape(rows, -1), x.reshape(rows, -1)
        g = _frame_chunks(rows)
        dy_c = dy_r.view(g, rows // g, -1)
        if weight.size(0) > 256:            # wide outputs fill the chip as one product
            dweight = dy_r.t().mm(x_r)
        else:
            dweight = _native.sum_leading(torch.bmm(dy_c.transpose(1, 2), x_r.view(g, rows // g, -1)))
        dbias = dy_c.sum(1).sum(0) if ctx.with_bias else None
        return dy_r.matmul(weight).view_as(x), dweight, dbias


_PROJ_KPAD = int(os.environ.get('ASR_PROJ_KPAD', '64'))     # extra K columns of the forward product (>= 2, multiple of 8)


_WIDE_DW_CHUNKS = int(os.environ.get('ASR_PROJ_DW_CHUNKS', '16'))


def _mm32(a, b):
    """bf16 x bf16 -> fp32 library product (MFMA, fp32 accumulation)"""
    return torch.mm(a, b, out_dtype=torch.float32)


class _WideFrameProjection(torch.autograd.Function):
    """The same product for a WIDE class layer (bi-character alphabets: C = 2401) as
    split-bf16 MFMA products.  In fp32 the three products of a step (logits, dx, dW: 0.79
    TFLOP at 171 k frames) run at the fp32 matrix rate and are a third of the bi-char step;
    with x = xh + xl, W = Wh + Wl (bf16 halves, `_native.split_bf16`),
        x Wt  ~  xh Wht + xh Wlt + xl Wht            (the xl Wlt term is 2^-16 of a 2^-8 term)
    runs at the bf16 rate with fp32 accumulation and keeps 2^-16 relative error per product —
    two orders below what the decoder's 1e-4 loss bound needs (tests/test_model_gpu.py holds
    the C = 2401 steps to the fp32 CPU reference).  Forward: ONE product over the
    K-concatenated operands [xh | xh | xl] [Wh | Wl | Wh]t (the sum happens in the MFMA
    accumulator); backward: three products each for dx and dW, summed in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y, saved, meta = _wide_forward(x, weight, bias)
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        return y

    @staticmethod
    def backward(ctx, dy):
        a, whp, wlp = ctx.saved_tensors
        with_bias, xshape, C, dgrad = ctx.meta
        Cp, rows = whp.size(0), a.size(0)
        dy_r = dy.reshape(rows, C)
        dbias = None
        if with_bias:
            g = _frame_chunks(rows)
            dbias = dy_r.view(g, rows // g, -1).sum(1).sum(0)
        dh = torch.empty((rows, Cp), dtype=torch.bfloat16, device=dy.device)
        dl = torch.empty((rows, Cp), dtype=torch.bfloat16, device=dy.device)
        if Cp > C:
            dh[:, C:] = 0.0
            dl[:, C:] = 0.0
        _native.split_bf16(dy_r, dh[:, :C], dl[:, :C])
        dx, dweight = _wide_backward(a, whp, wlp, dh, dl, C, dgrad)
        return dx.view(xshape), dweight, dbias


def _wide_forward(x, weight, bias):
    """-> (logits [..., C] f32, tensors to save, (with_bias, x.shape, C, dgrad))"""
    rows, K = x.numel() // x.size(-1), x.size(-1)
    C = weight.size(0)
    # (the backward's dx products run on csrc/lstm_dgrad.hip when the shapes are the ones
    # it is built for — [rows, 2560] x [2560, 320]: 2.6x the library's pick for them)
    dgrad = K == 320 and C <= 2560 and _native.lstm_dgrad_supported(K) and rows * 5120 < 2 ** 31 - 2 ** 20
    Cp = 2560 if dgrad else (C + 63) // 64 * 64
    Kc = 3 * K + _PROJ_KPAD
    x_r = x.reshape(rows, K)
    # [xh | xh | xl | 1 1 0..0] x [Wh | Wl | Wh | bh bl 0..0]t: the bias rides in the product
    # (a separate `y += b` is one more pass over the 1.6 GB of logits); the extra columns
    # keep the rows 16-byte aligned for the library's fast kernels
    a = torch.empty((rows, Kc), dtype=torch.bfloat16, device=x.device)
    _native.split_bf16(x_r, a[:, :K], a[:, 2 * K:3 * K])
    a[:, K:2 * K].copy_(a[:, :K])
    a[:, 3 * K:3 * K + 2] = 1.0
    a[:, 3 * K + 2:] = 0.0
    # the halves of W, zero-padded to Cp rows: the backward products run over Cp columns of
    # dy — rows of 2401 bf16 are not 16-byte aligned and send the library to kernels three
    # times slower
    whp = torch.zeros((Cp, K), dtype=torch.bfloat16, device=x.device)
    wlp = torch.zeros((Cp, K), dtype=torch.bfloat16, device=x.device)
    _native.split_bf16(weight.detach(), whp[:C], wlp[:C])
    wcat = torch.zeros((C, Kc), dtype=torch.bfloat16, device=x.device)
    wcat[:, :K] = whp[:C]
    wcat[:, K:2 * K] = wlp[:C]
    wcat[:, 2 * K:3 * K] = whp[:C]
    if bias is not None:
        bh, bl = _native.split_bf16(bias.detach())
        wcat[:, 3 * K] = bh
        wcat[:, 3 * K + 1] = bl
    y = _mm32(a, wcat.t())
    return y.view(x.shape[:-1] + (C,)), (a, whp, wlp), (bias is not None, x.shape, C, dgrad)


def _wide_backward(a, whp, wlp, dh, dl, C, dgrad):
    """dx [rows, K], dW [C, K] from the halves dh + dl [rows, Cp] of the logits' gradient"""
    Cp, K = whp.shape
    rows = a.size(0)
    xh, xl = a[:, :K], a[:, 2 * K:3 * K]
    if dgrad:
        def mm(d, w):
            return _native.lstm_dgrad(d.view(rows, 1, 2, Cp // 2), w).view(rows, K)
    else:
        mm = _mm32
    dx = mm(dh, whp)
    dx += mm(dh, wlp)
    dx += mm(dl, whp)
    # dW: a product over ALL frames into a small [C, K] output — as one library call it runs on
    # ~20 workgroups (0.8 ms per term); split over G chunks of frames and summed
    G = _WIDE_DW_CHUNKS if rows % _WIDE_DW_CHUNKS == 0 else 1
    dh3, dl3 = dh.view(G, rows // G, Cp).transpose(1, 2), dl.view(G, rows // G, Cp).transpose(1, 2)
    xh3, xl3 = xh.reshape(G, rows // G, K), xl.reshape(G, rows // G, K)
    part = torch.bmm(dh3, xh3, out_dtype=torch.float32)
    part += torch.bmm(dh3, xl3, out_dtype=torch.float32)
    part += torch.bmm(dl3, xh3, out_dtype=torch.float32)
    dweight = _native.sum_leading(part) if G > 1 else part[0]
    return dx, dweight[:C]


class _WideProjectNormaliseShift(torch.autograd.Function):
    """_WideFrameProjection followed by _NormaliseShift (FSTDecoder with normalize_by_dim = 0)
    as ONE autograd node: the gradient of the logits never exists in fp32 —
    asr_log_softmax_shift_bwd_split_bf16 writes its bf16 halves and its column sums (the bias
    gradient) directly: one 1.6 GB write and the split pass's read of it less per step."""

    @staticmethod
    def forward(ctx, x, weight, bias, lens_dev):
        y, saved, meta = _wide_forward(x, weight, bias)
        shifted, nls, nls_sum = _native.log_softmax_shift_fwd(y, lens_dev)
        ctx.save_for_backward(shifted, nls, *saved)
        ctx.meta = meta
        ctx.mark_non_differentiable(nls_sum)
        return shifted, nls_sum

    @staticmethod
    def backward(ctx, dshifted, _):
        shifted, nls, a, whp, wlp = ctx.saved_tensors
        with_bias, xshape, C, dgrad = ctx.meta
        dh, dl, dbias = _native.log_softmax_shift_bwd_split(shifted, nls, dshifted.contiguous(), whp.size(0))
        dx, dweight = _wide_backward(a, whp, wlp, dh, dl, C, dgrad)
        return dx.view(xshape), dweight, (dbias if with_bias else None), None


def _wide_applies(frames, weight):
    return (frames.is_cuda and frames.dim() >= 2 and frames.numel() // frames.size(-1) >= 4096
            and weight.size(0) > 256 and frames.dtype == torch.float32 and weight.dtype == torch.float32
            and os.environ.get('ASR_PROJ_SPLIT', '1') != '0')


def project_normalise_shift(layer, frames, lens_dev):
    """`_NormaliseShift(project_frames(layer, frames), lens)`; one node for wide alphabets"""
    weight, bias = layer.class_weight_bias()
    if _wide_applies(frames, weight) and weight.size(0) <= 2560 and frames.dim() == 3:
        if layer.training:
            logger.log_scalar("ngram_linear_weight_norm", torch.norm(weight))
        return _WideProjectNormaliseShift.apply(frames, weight, bias, lens_dev)
    return None


def project_frames(layer, frames):
    """Apply a class-projection layer (`class_weight_bias()`) to `[..., F]` frames."""
    weight, bias = layer.class_weight_bias()
    if layer.training:
        logger.log_scalar("ngram_linear_weight_norm", torch.norm(weight))
    many = frames.dim() >= 2 and frames.numel() // frames.size(-1) >= 4096
    if frames.is_cuda and many:
        if _wide_applies(frames, weight):        # ASR_PROJ_SPLIT=0: the fp32 products (A/B runs)
            return _WideFrameProjection.apply(frames, weight, bias)
        return _FrameProjection.apply(frames, weight, bias)
    return F.linear(frames, weight, bias)


def _with_training_noise(layer, weight):
    # `weight_noise` is driven by the training hook (hooks/weight_noise.py:41-52)
    if layer.training and layer.weight_noise > 0:
        return weight + layer.weight_noise * torch.randn_like(weight)
    return weight


class LutLinear(nn.Linear):
    """One learned prototype row per output class (reference :29-70).  With
    `tie_blanks` every class whose last symbol is the blank uses row 0."""

    def __init__(self, in_dim, num_symbols, ngram_to_class, bias=True,
                 tie_blanks=False, bias_only_for_dim=None):
        num_classes = ngram_to_class.size(0)
        super(LutLinear, self).__init__(in_dim, num_classes, bias)
        self.num_symbols = num_symbols
        self.weight_noise = 0.0
        self.tied_w_rows = None
        if tie_blanks:
            rows = torch.arange(num_classes, dtype=torch.long)
            rows[::num_symbols] = 0
            self.tied_w_rows = rows

    def class_weight_bias(self):
        weight, bias = _with_training_noise(self, self.weight), self.bias
        if self.tied_w_rows is not None:
            rows = self.tied_w_rows.to(weight.device)
            weight = weight.index_select(0, rows)
            bias = None if bias is None else bias.index_select(0, rows)
        return weight, bias

    def forward(self, input):
        return project_frames(self, input)


class GatedAct(nn.Module):
    """tanh half gated by the sigmoid of the other half of the last axis."""

    def forward(self, x):
        gate, value = x.chunk(2, dim=-1)
        return torch.tanh(value) * torch.sigmoid(gate)


_NGRAM_ACTIVATIONS = {'relu': nn.ReLU, 'tanh': nn.Tanh, 'gated': GatedAct}
_NGRAM_COMBINE = ('sum', 'concat', 'lstm')


class NGramLinear(nn.Module):
    """Class prototypes COMPUTED from symbol embeddings (reference :79-223): the
    n-gram of a class is embedded symbol by symbol, the embeddings are combined
    (`sum`, `concat`, or the last state of an `lstm` over them) and pushed through
    `num_layers` dense layers; the result is the `[C, in_dim]` projection matrix.

    Parameter names (checkpoint keys): `embedding.weight`, `weight_computer.<i>.*`
    (dense layers at the even positions of the stack, or shifted by the Dropout
    modules when `dropout > 0`), `weight_computer_.*` for the lstm, `bias`."""

    def __init__(self, in_dim, num_symbols, ngram_to_class, bias=True,
                 bias_only_for_dim=None, inner_dim=None, dropout=0.0,
                 embedding_dim=None, tied_embeddings=True,
                 embedding_combination_method='sum',
                 num_layers=0, weight_noise=0.0, activation='relu'):
        super(NGramLinear, self).__init__()
        if embedding_combination_method not in _NGRAM_COMBINE:
            raise ValueError("Unknown embedding_combination_method")
        assert embedding_combination_method != 'lstm' or num_layers <= 1
        table = ngram_to_class.clone().long()
        num_classes, order = table.shape
        pieces = order if embedding_combination_method == 'concat' else 1
        hidden = inner_dim or in_dim
        self.num_symbols, self.in_dim, self.inner_dim = num_symbols, in_dim, hidden
        self.embedding_dim = embedding_dim or hidden // pieces
        self.embedding_combination_method = embedding_combination_method
        self.tied_embeddings, self.num_layers = tied_embeddings, num_layers
        self.dropout, self.weight_noise = dropout, 0.0
        self.bias_only_for_dim = bias_only_for_dim

        # symbol table: one row per symbol, or per (n-gram position, symbol)
        positions = 1 if tied_embeddings else order
        if not tied_embeddings:
            table = table + num_symbols * torch.arange(order, dtype=torch.long)[None, :]
        self.embedding = nn.Embedding(num_symbols * positions, self.embedding_dim)
        # buffers follow .to(device); non-persistent because the reference keeps plain
        # attributes here, i.e. no state_dict entries
        self.register_buffer('ngram_to_class', table, persistent=False)

        if not bias:
            self.register_parameter('bias', None)
        elif bias_only_for_dim is None:
            self.bias = nn.Parameter(torch.zeros(num_classes))
        else:           # one bias per symbol at one n-gram position
            self.bias = nn.Parameter(torch.zeros(num_symbols, 1))
            self.register_buffer('ngram_to_bias',
                                 ngram_to_class[:, bias_only_for_dim].clone().long(),
                                 persistent=False)

        self._build_prototype_net(pieces * self.embedding_dim, activation)
        self._cache = None

    def _build_prototype_net(self, first_in, activation):
        n = self.num_layers
        if n == 0:
            if self.embedding_combination_method == 'lstm':
                self.weight_computer_ = nn.LSTM(self.in_dim, self.in_dim, num_layers=1)
            return
        act = _NGRAM_ACTIVATIONS[activation]
        # widths the reference allocates: hidden layers `inner_dim` (doubled for the
        # gated activation), the last one `in_dim`; a layer's input is its predecessor's
        # allocated width
        widths = [self.inner_dim * (2 if activation == 'gated' else 1)] * (n - 1) + [self.in_dim]
        stack = [nn.Linear(first_in, widths[0])]
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            stack += [act(), nn.Linear(fan_in, fan_out)]
        if self.dropout:        # behind the first dense layer and behind every activation
            first = stack[0]
            stack = [m for layer in stack for m in (
                (layer, nn.Dropout(self.dropout))
                if layer is first or not isinstance(layer, nn.Linear) else (layer,))]
        self.weight_computer = nn.Sequential(*stack)

    # -- prototypes ---------------------------------------------------------
    def _combined_embeddings(self):
        e = self.embedding(self.ngram_to_class)                  # [C, order, D]
        how = self.embedding_combination_method
        if how == 'sum':
            return e.sum(1)
        if how == 'concat':
            return e.flatten(1)
        return e.transpose(0, 1)                                 # lstm: [order, C, D]

    def compute_weight(self):
        z = self._combined_embeddings()
        if self.num_layers > 0:
            return self.weight_computer(z)
        if self.embedding_combination_method == 'lstm':
            return self.weight_computer_(z)[0][-1]
        return z

    def class_weight_bias(self):
        """Prototype matrix and bias of this step.  Without autograd (decoding, scoring)
        the network's output is kept while no parameter has changed and nothing random
        sits in it, so the passes of one evaluation compute it once."""
        reusable = not torch.is_grad_enabled() and not (self.training and self.dropout)
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) if reusable else None
        if reusable and self._cache is not None and self._cache[0] == key:
            weight = self._cache[1]
        else:
            weight = self.compute_weight()
            self._cache = (key, weight) if reusable else None
        bias = self.bias
        if bias is not None and self.bias_only_for_dim is not None:
            bias = bias[self.ngram_to_bias, 0]
        return _with_training_noise(self, weight), bias

    def forward(self, input):
        return project_frames(self, input)


# ---------------------------------------------------------------------------
# decoders
# ---------------------------------------------------------------------------
class _ProjectionDecoder(BaseDecoder):
    """What CTCDecoderAdvanced and FSTDecoder have in common: `fc` (the embedder
    under the reference's module path `fc.0.module.0`) and the optional entries
    of the dict `decode` returns."""

    def _make_fc(self, sample_batch, embedder, embedder_kwargs, num_symbols, ngram_to_class):
        encoded_dim = sample_batch["features"].size(2)
        layer = globals()[embedder](encoded_dim, num_symbols, ngram_to_class, **embedder_kwargs)
        self.fc = nn.Sequential(SequenceWise(nn.Sequential(layer)))

    @property
    def embedder(self):
        return self.fc[0].module[0]

    def _utterance_losses(self, logits, lens, texts, text_lens, other):
        raise NotImplementedError

    def _finish_decode(self, ret, loss_key, logits, lens, texts, text_lens, other,
                       with_generated, with_length_diff, empty_other):
        """`loss` for known transcripts; `text_loss` / `generated_loss`: per-utterance
        losses of the transcripts and of the decoder's own output; `logits_text_diff`:
        frames minus labels."""
        text_losses = None
        if texts is not None and text_lens is not None:
            text_losses = self._utterance_losses(logits, lens, texts, text_lens, other)
            total = text_losses.sum()
            ret['loss'] = {loss_key: total, 'loss': total}
        if with_generated:
            own = ret['decoded']
            own_lens = torch.IntTensor([len(seq) for seq in own])
            ret['text_loss'] = text_losses.tolist()
            ret['generated_loss'] = self._utterance_losses(
                logits, lens, own, own_lens, empty_other).tolist()
        if with_length_diff:
            ret['logits_text_diff'] = (torch.as_tensor(lens) - torch.as_tensor(text_lens)).tolist()
        return ret


class CTCDecoderAdvanced(_ProjectionDecoder):
    """Frame-synchronous CTC decoder with a pluggable loss function and greedy
    decoding (reference :226-391)."""

    def __init__(self, sample_batch, num_classes, context_order=1,
                 normalize_by_dim=None, ctc_loss_fn='ctc_loss',
                 ctc_allow_nonblank_selfloops=True,
                 loop_using_symbol_repetitions=False,
                 embedder='LutLinear', embedder_kwargs={},
                 bigram_dovetail_decoder=False,
                 local_normalization=True,
                 fix_greedy_decoder=False,
                 **kwargs):
        super(CTCDecoderAdvanced, self).__init__(**kwargs)
        if bigram_dovetail_decoder:
            raise NotImplementedError("bigram_dovetail_decoder is unused by the shipped configs")
        self.ctc_loss_fn = globals()[ctc_loss_fn]
        symbols = int(round(num_classes ** (1.0 / context_order)))
        assert symbols ** context_order == num_classes and symbols == self.num_symbols
        table = fst_utils.make_full_ngram_table(context_order, symbols, num_classes)[2]
        self.blanks = list(range(0, num_classes, symbols))      # classes ending in the blank
        self.context_order, self.normalize_by_dim = context_order, normalize_by_dim
        self.ctc_allow_nonblank_selfloops = ctc_allow_nonblank_selfloops
        self.loop_using_symbol_repetitions = loop_using_symbol_repetitions
        self.bigram_dovetail_decoder = False
        self.local_normalization = local_normalization
        self.fix_greedy_decoder = fix_greedy_decoder
        self._make_fc(sample_batch, embedder, embedder_kwargs, symbols, table)

    def logits(self, encoded, encoded_lens=None, normalize_logits=True):
        acts = self.fc(encoded)
        if not self.local_normalization:
            return acts
        return get_normalized_acts(acts, encoded_lens, self.num_symbols, self.context_order,
                                   self.normalize_by_dim, normalize_logits)

    def get_ctc_losses(self, logits, logit_lens, texts, text_lens, other_data_in_batch):
        lens = [int(l) for l in torch.as_tensor(text_lens).tolist()]
        flat = torch.cat([torch.as_tensor(t)[:l] for t, l in zip(texts, lens)])
        # ten POSITIONAL arguments, as the reference passes them (:292-297): slot ten
        # means eval_repeats_in_context to ctc_loss and other_data_in_batch to ctc_fst_loss
        return self.ctc_loss_fn(logits, flat, logit_lens, text_lens, self.num_symbols,
                                self.context_order, self.normalize_by_dim,
                                self.ctc_allow_nonblank_selfloops,
                                self.loop_using_symbol_repetitions, other_data_in_batch)

    _utterance_losses = get_ctc_losses

    def forward(self, encoded, encoded_lens, texts, text_lens, spkids=None,
                **other_data_in_batch):
        total = self.get_ctc_losses(self.fc(encoded), encoded_lens, texts, text_lens,
                                    other_data_in_batch).sum()
        return {'ctc_loss': total, 'loss': total}

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_texts_and_generated_loss=False,
               return_logits_text_diff=False, spkids=None,
               **other_data_in_batch):
        logits = self.logits(encoded, encoded_lens)
        best = _native.argmax_rows(logits.detach()).t().cpu().long()      # [B, T'] (:325,352)
        ret = {'decoded': self.process_sequences(best, encoded_lens),
               'decoded_frames': best, 'logits': logits}
        return self._finish_decode(ret, 'ctc_loss', logits, encoded_lens, texts, text_lens,
                                   other_data_in_batch, return_texts_and_generated_loss,
                                   return_logits_text_diff, {})

    def process_sequences(self, frames, frame_lens):
        return [self.process_sequence(frames[b], frame_lens[b]) for b in range(len(frame_lens))]

    def process_sequence(self, frames, num_frames):
        """Greedy collapse, bug-compatible with the reference's default branch
        (:385-391).  Its adjacent-frame test `i != 0 or c != frames[i-1]` is always true
        past frame 0, and at frame 0 compares with the LAST element of the padded row;
        what actually removes repeats is the comparison (mod num_symbols) with the
        previously KEPT class."""
        if self.fix_greedy_decoder:
            # the reference calls a function that does not exist (:381)
            raise NameError("name 'remove_repetitions_blanks' is not defined")
        row = frames.tolist()
        S, blank_classes, kept = self.num_symbols, frozenset(self.blanks), []
        for i, c in enumerate(row[:int(num_frames)]):
            c = int(c)
            if c in blank_classes or (i == 0 and c == int(row[-1])):
                continue
            if kept and kept[-1] % S == c % S:
                continue
            kept.append(c)
        return kept


class _SubRowMax(torch.autograd.Function):
    """logits - max_c(logits) and sum_t max_t * [t < len] in one pass (reference
    :479-484).  The reference detaches the maximum, so the gradient passes through."""

    @staticmethod
    def forward(ctx, logits, lens_dev):
        shifted, _, max_sum = _native.sub_rowmax(logits.contiguous(), lens_dev)
        ctx.mark_non_differentiable(max_sum)
        return shifted, max_sum

    @staticmethod
    def backward(ctx, dshifted, _):
        return dshifted, None


class _NormaliseShift(torch.autograd.Function):
    """get_normalized_acts(normalize_by_dim=0) followed by _SubRowMax in one pass over the
    logits (asr_log_softmax_shift_*): the normaliser cancels in the shifted acts and only
    enters the per-utterance sum of row maxima."""

    @staticmethod
    def forward(ctx, acts, lens_dev):
        shifted, nls, nls_sum = _native.log_softmax_shift_fwd(acts.contiguous(), lens_dev)
        ctx.save_for_backward(shifted, nls)
        ctx.mark_non_differentiable(nls_sum)
        return shifted, nls_sum

    @staticmethod
    def backward(ctx, dshifted, _):
        shifted, nls = ctx.saved_tensors
        return _native.log_softmax_shift_bwd(shifted, nls, dshifted.contiguous()), None


class FSTDecoder(_ProjectionDecoder):
    """Lattice-based decoder: loss = numerator reduction over the utterance's
    training graph minus a denominator (reduction over the decoding graph, or the
    subtracted row maxima when the acts are locally normalised); decoding = best
    path through the decoding graph (reference :394-593)."""

    def __init__(self, sample_batch, num_classes,
                 graph_generator, normalize_by_dim=None,
                 numerator_red='logsumexp', denominator_red='logsumexp',
                 embedder='LutLinear', embedder_kwargs={}, **kwargs):
        super(FSTDecoder, self).__init__(**kwargs)
        gg = utils.contruct_from_kwargs(
            graph_generator, 'att_speech.fst_utils',
            {'num_classes': num_classes, 'num_symbols': self.num_symbols})
        self.graph_generator, self.context_order = gg, gg.context_order
        self.normalize_by_dim = normalize_by_dim
        self.numerator_red, self.denominator_red = numerator_red, denominator_red
        self.verbose = False
        self._verify = False              # batch graphs checked against our own once
        if normalize_by_dim not in (None, 0):
            assert gg.num_classes == gg.num_symbols ** gg.context_order
        if self.num_symbols is None:
            self.num_symbols = gg.num_symbols
        self.dec_fst = gg.decoding_fst    # transducer that maps a state path to labels
        self._make_fc(sample_batch, embedder, embedder_kwargs, gg.num_symbols, gg.ngram_to_class)

    def logits(self, encoded, encoded_lens=None, extra_ret=None):
        acts = self.fc(encoded)
        if extra_ret is not None:
            extra_ret['unnormed_logits'] = acts
        if self.normalize_by_dim is None:
            return acts
        return get_normalized_acts(acts, encoded_lens, self.num_symbols, self.context_order,
                                   self.normalize_by_dim, normalize_logits=True)

    def _numerator_graphs(self, texts, text_lens, other, device):
        given = other.get('graph_matrices') if other else None
        if isinstance(given, _native.Graph):      # built on the device at the start of the step
            return given
        if given is None:
            # built on the device from the labels (the reference builds them on the
            # host at this point, :470-471, or in its data workers)
            return self.graph_generator.get_training_graph_device(texts, text_lens, device)
        if not self._verify:              # first batch: the data pipeline's == ours (:460-468)
            ours = self.graph_generator.get_training_matrices_batch(texts, text_lens, 'cpu')
            worst = max(float((a.cpu() - b).abs().max()) for a, b in zip(given, ours))
            assert worst < 1e-10
            self._verify = True
        return given

    def get_fst_loss(self, logits, encoded_lens, texts, text_lens, other_data_in_batch,
                     unnormalised=False):
        """`unnormalised`: `logits` are the raw acts of a decoder with normalize_by_dim = 0;
        normalisation and stabilisation then run as one pass"""
        gg = self.graph_generator
        device = logits[0].device if isinstance(logits, tuple) else logits.device
        numerator = self._numerator_graphs(texts, text_lens, other_data_in_batch, device)
        lens_dev = _native.lens_on(encoded_lens, device)
        if isinstance(logits, tuple):                 # (shifted, max_sum) of project_normalise_shift
            shifted, max_sum = logits
        elif unnormalised:
            shifted, max_sum = _NormaliseShift.apply(logits, lens_dev)   # (:444-452) + (:479-484)
        else:
            shifted, max_sum = _SubRowMax.apply(logits, lens_dev)        # (:479-484)
        if self.denominator_red in ('logsumexp', 'logsumexp_fwb') and \
                self.numerator_red in ('logsumexp', 'logsumexp_fwb') and shifted.is_cuda and \
                not self.verbose and os.environ.get('ASR_FUSED_NUMDEN', '1') != '0':
            # both reductions in one autograd node: one gradient buffer, no [T,B,C] addition
            den_graph = gg.get_decoding_matrices('cpu')
            grouped = fst_utils._device_grouped(den_graph, shifted.device)
            num_ok = isinstance(numerator, _native.Graph) or len(numerator) == 8
            if grouped is not None and num_ok:
                return fst_utils.NumeratorMinusDenominator.apply(
                    shifted, encoded_lens, numerator, grouped, gg.nc_weight)[0]
        num = fst_utils.path_reduction(shifted, encoded_lens, numerator, red_kind=self.numerator_red,
                                       neg_inf=gg.nc_weight, negate=True)
        if self.denominator_red == 'none':
            den = max_sum
        else:
            den = -fst_utils.path_reduction(shifted, encoded_lens, gg.get_decoding_matrices('cpu'),
                                            red_kind=self.denominator_red, neg_inf=gg.nc_weight)
        if self.verbose:
            print("global loss: [loss: num %g, den %g, com %g]" % (
                float(num.sum()), float(den.sum()), -float(max_sum.sum())))
        return num - den

    _utterance_losses = get_fst_loss

    def forward(self, encoded, encoded_lens, texts, text_lens, spkids=None,
                **other_data_in_batch):
        if self.normalize_by_dim == 0 and self.normalize_by_dim is not None and encoded.is_cuda:
            # wide alphabets: projection + normalisation + stabilisation as one autograd node
            fused = project_normalise_shift(self.embedder, encoded,
                                            _native.lens_on(encoded_lens, encoded.device))
            losses = self.get_fst_loss(fused if fused is not None else self.fc(encoded), encoded_lens,
                                       texts, text_lens, other_data_in_batch, unnormalised=True)
        else:
            losses = self.get_fst_loss(self.logits(encoded, encoded_lens), encoded_lens,
                                       texts, text_lens, other_data_in_batch)
        total = losses.sum()
        return {'fst_loss': total, 'loss': total}

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_texts_and_generated_loss=False,
               return_logits_text_diff=False, spkids=None,
               **other_data_in_batch):
        logits = self.logits(encoded, encoded_lens)
        gg = self.graph_generator
        # best state sequence; the reference reads the same indices off the autograd
        # gradient of the Viterbi score (:546-554)
        states = fst_utils.viterbi_path(logits.detach(), encoded_lens,
                                        gg.get_decoding_matrices('cpu'), gg.nc_weight)[1]
        states = states.cpu().numpy()
        frames = [int(l) for l in torch.as_tensor(encoded_lens).tolist()]
        ret = {'decoded': [self.dec_fst.read_out(states[:n, b]) for b, n in enumerate(frames)],
               'logits': logits}                                         # (:556-571)
        return self._finish_decode(ret, 'fst_loss', logits, encoded_lens, texts, text_lens,
                                   other_data_in_batch, return_texts_and_generated_loss,
                                   return_logits_text_diff, None)
