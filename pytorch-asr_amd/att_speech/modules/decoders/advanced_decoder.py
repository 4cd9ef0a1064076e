"""att_speech.modules.decoders.advanced_decoder — MI355X-native counterpart of
the reference module of the same dotted name
(att_speech/modules/decoders/advanced_decoder.py): LutLinear (:29-70),
NGramLinear (:79-223), CTCDecoderAdvanced (:226-391), FSTDecoder (:394-593).
Constructor kwargs, forward/decode signatures, return dicts and state_dict keys
are the reference's; the lattice, normalisation and Viterbi
arithmetic runs in the HIP kernels behind include/asr_amd.h."""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from att_speech import _native, fst_utils, utils
from att_speech.configuration import Globals
from att_speech.logger import DefaultTensorLogger
from att_speech.modules.common import SequenceWise
from att_speech.modules.ctc_losses import (  # noqa: F401
    ctc_fst_loss, ctc_loss, get_normalized_acts)
from att_speech.modules.decoders.base_decoder import BaseDecoder

logger = DefaultTensorLogger()


class _SkinnyLinear(torch.autograd.Function):
    """y = x W^T + b for a SKINNY projection (few classes, very many frames) on the
    GPU.  Forward is the library GEMM; the two reductions over the frames that the
    backward needs are reshaped so they fill the chip: torch's dW GEMM gets 49
    workgroups for a [C x frames] x [frames x 320] product (0.5 ms at 171 k frames)
    and its bias reduction ONE 64-thread workgroup (0.86 ms); here both are
    split over G chunks of frames (batched GEMM / two-stage sum)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.size(-1))
        x2 = x.reshape(-1, x.size(-1))
        rows = dy2.size(0)
        dx = dy2.matmul(w).view_as(x)
        g = 1
        for cand in (256, 128, 64, 32, 16, 8, 4, 2):
            if rows % cand == 0 and rows // cand >= 64:
                g = cand
                break
        dyc = dy2.view(g, rows // g, -1)
        if w.size(0) <= 256:
            dw = torch.bmm(dyc.transpose(1, 2), x2.view(g, rows // g, -1)).sum(0)
        else:                       # wide outputs (bigram classes) fill the chip as one GEMM
            dw = dy2.t().mm(x2)
        db = dyc.sum(1).sum(0) if ctx.has_bias else None
        return dx, dw, db


class LutLinear(nn.Linear):
    """Look-up table for softmax, one prototype per class (reference :29-70)."""

    def __init__(self, in_dim, num_symbols, ngram_to_class, bias=True,
                 tie_blanks=False, bias_only_for_dim=None):
        self.num_symbols = num_symbols
        if tie_blanks:
            tied_w_rows = torch.arange(ngram_to_class.size(0), dtype=torch.long)
            tied_w_rows[::num_symbols] = 0
            self.tied_w_rows = tied_w_rows
        else:
            self.tied_w_rows = None
        # set by the weight-noise training hook (hooks/weight_noise.py:41-52)
        self.weight_noise = 0.0
        super(LutLinear, self).__init__(in_dim, ngram_to_class.size(0), bias)

    def forward(self, input):
        w = self.weight
        if self.training:
            logger.log_scalar("ngram_linear_weight_norm", torch.norm(w))
        if self.training and self.weight_noise > 0:
            w = w + torch.randn_like(w) * self.weight_noise
        b = self.bias
        if self.tied_w_rows is not None:
            rows = self.tied_w_rows.to(w.device)
            w = w[rows]
            b = b[rows]
        if input.is_cuda and input.dim() >= 2 and input.numel() // input.size(-1) >= 4096:
            return _SkinnyLinear.apply(input, w, b)
        return F.linear(input, w, b)


class GatedAct(nn.Module):
    def forward(self, x):
        g, x = torch.chunk(x, 2, dim=-1)
        return torch.sigmoid(g) * torch.tanh(x)


class NGramLinear(nn.Module):
    """Tied prototype mapping for softmax: the [C, in_dim] projection weight is
    COMPUTED from symbol embeddings every step (reference :79-223)."""

    def __init__(self, in_dim, num_symbols, ngram_to_class, bias=True,
                 bias_only_for_dim=None, inner_dim=None, dropout=0.0,
                 embedding_dim=None, tied_embeddings=True,
                 embedding_combination_method='sum',
                 num_layers=0, weight_noise=0.0, activation='relu'):
        super(NGramLinear, self).__init__()
        self.num_symbols = num_symbols
        ngram_to_class = ngram_to_class.clone()
        num_classes, ngram_order = ngram_to_class.size()
        self.in_dim = in_dim
        self.inner_dim = inner_dim or in_dim
        self.dropout = dropout
        embedding_multiplier = {'concat': ngram_order, 'lstm': 1,
                                'sum': 1}[embedding_combination_method]
        assert not (embedding_combination_method == 'lstm' and num_layers > 1)
        embedding_dim = embedding_dim or self.inner_dim // embedding_multiplier
        activation_class = {'tanh': nn.Tanh, 'relu': nn.ReLU,
                            'gated': GatedAct}[activation]

        def layer_dim(lnum):
            if lnum == num_layers - 1:
                return self.in_dim
            ret_dim = self.inner_dim
            if activation == 'gated':
                ret_dim *= 2
            return ret_dim

        self.embedding_dim = embedding_dim
        self.tied_embeddings = tied_embeddings
        self.embedding_combination_method = embedding_combination_method
        self.num_layers = num_layers
        self.weight_noise = 0.0
        if bias:
            self.bias_only_for_dim = bias_only_for_dim
            if bias_only_for_dim is None:
                self.bias = nn.Parameter(torch.zeros(num_classes))
            else:
                self.bias = nn.Parameter(torch.zeros(num_symbols, 1))
                self.register_buffer('ngram_to_bias',
                                     ngram_to_class[:, bias_only_for_dim].clone(),
                                     persistent=False)
        else:
            self.register_parameter('bias', None)
        if tied_embeddings:
            num_embeddings = self.num_symbols
        else:
            num_embeddings = self.num_symbols * ngram_order
            shift = torch.arange(ngram_order, dtype=torch.long).view(1, -1) * num_symbols
            ngram_to_class = ngram_to_class + shift.to(ngram_to_class.device)
        # a buffer so it follows .to(device); not persistent: the reference keeps
        # it as a plain attribute, so it is not a state_dict key there either
        self.register_buffer('ngram_to_class', ngram_to_class, persistent=False)
        self.embedding = torch.nn.Embedding(num_embeddings, embedding_dim)
        if num_layers > 0:
            net_input_dim = embedding_multiplier * embedding_dim
            layers = [nn.Linear(net_input_dim, layer_dim(0))]
            if dropout:
                layers.append(nn.Dropout(dropout))
            for lnum in range(1, num_layers):
                layers.append(activation_class())
                if dropout:
                    layers.append(nn.Dropout(dropout))
                layers.append(nn.Linear(layer_dim(lnum - 1), layer_dim(lnum)))
            self.weight_computer = nn.Sequential(*layers)
        elif self.embedding_combination_method == 'lstm':
            self.weight_computer_ = nn.LSTM(self.in_dim, self.in_dim, num_layers=1)
            self.weight_computer = lambda x: self.weight_computer_(x)[0][-1]
        else:
            self.weight_computer = lambda x: x

    def compute_weight(self):
        embedded = self.embedding(self.ngram_to_class)
        if self.embedding_combination_method == 'concat':
            embedded = embedded.view(embedded.size(0), -1)
        elif self.embedding_combination_method == 'sum':
            embedded = embedded.sum(1)
        elif self.embedding_combination_method == 'lstm':
            embedded = embedded.transpose(0, 1)
        else:
            raise ValueError("Unknown embedding_combination_method")
        return self.weight_computer(embedded)

    def forward(self, input):
        weight = self.compute_weight()
        if self.training:
            logger.log_scalar("ngram_linear_weight_norm", torch.norm(weight))
        if self.training and self.weight_noise > 0:
            weight = weight + torch.randn_like(weight) * self.weight_noise
        if self.bias is not None:
            if self.bias_only_for_dim is None:
                bias = self.bias
            else:
                bias = F.embedding(self.ngram_to_bias, self.bias).view(-1)
        else:
            bias = None
        return F.linear(input, weight, bias)


def _make_ngram_table(context_order, num_classes):
    num_symbols = int(round(num_classes ** (1.0 / context_order)))
    assert num_symbols ** context_order == num_classes
    _, _, ngram_to_class = fst_utils.make_full_ngram_table(
        context_order, num_symbols, num_classes)
    blanks = [i for i in range(num_classes) if i % num_symbols == 0]
    return num_symbols, ngram_to_class, blanks


class CTCDecoderAdvanced(BaseDecoder):
    """reference :226-391"""

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
        self.ctc_loss_fn = globals()[ctc_loss_fn]
        num_symbols, ngram_to_class, blanks = _make_ngram_table(
            context_order, num_classes)
        assert self.num_symbols == num_symbols
        self.context_order = context_order
        self.normalize_by_dim = normalize_by_dim
        self.bigram_dovetail_decoder = bigram_dovetail_decoder
        self.blanks = blanks
        self.ctc_allow_nonblank_selfloops = ctc_allow_nonblank_selfloops
        self.loop_using_symbol_repetitions = loop_using_symbol_repetitions
        self.local_normalization = local_normalization
        self.fix_greedy_decoder = fix_greedy_decoder
        if bigram_dovetail_decoder:
            raise NotImplementedError("bigram_dovetail_decoder is unused by the shipped configs")
        rnn_hidden_size = sample_batch["features"].size()[2]
        embedder = globals()[embedder]
        modules = [embedder(rnn_hidden_size, num_symbols, ngram_to_class,
                            **embedder_kwargs)]
        fully_connected = nn.Sequential(*modules)
        self.fc = nn.Sequential(SequenceWise(fully_connected))

    def logits(self, encoded, encoded_lens=None, normalize_logits=True):
        logits = self.fc(encoded)
        if self.local_normalization:
            logits = get_normalized_acts(
                logits, encoded_lens, self.num_symbols, self.context_order,
                self.normalize_by_dim, normalize_logits)
        return logits

    def get_ctc_losses(self, logits, logit_lens, texts, text_lens,
                       other_data_in_batch):
        text_cat = torch.cat(
            [torch.as_tensor(t[:l]) for (t, l) in zip(texts, torch.as_tensor(text_lens))])
        # positional, like the reference (:292-297): the 10th slot differs
        # between ctc_loss and ctc_fst_loss
        return self.ctc_loss_fn(
            logits, text_cat, logit_lens, text_lens, self.num_symbols,
            self.context_order, self.normalize_by_dim,
            self.ctc_allow_nonblank_selfloops,
            self.loop_using_symbol_repetitions,
            other_data_in_batch)

    def forward(self, encoded, encoded_lens, texts, text_lens, spkids=None,
                **other_data_in_batch):
        unnormalised_logits = self.fc(encoded)
        ctc_loss = self.get_ctc_losses(
            unnormalised_logits, encoded_lens, texts, text_lens,
            other_data_in_batch).sum()
        return {'ctc_loss': ctc_loss, 'loss': ctc_loss}

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_texts_and_generated_loss=False,
               return_logits_text_diff=False, spkids=None,
               **other_data_in_batch):
        logits = self.logits(encoded, encoded_lens)
        ctc_loss = None
        if texts is not None and text_lens is not None:
            ctc_text_losses = self.get_ctc_losses(
                logits, encoded_lens, texts, text_lens, other_data_in_batch)
            ctc_loss = ctc_text_losses.sum()
        # per-frame arg-max over classes, [B, T'] on the host (:325,352)
        maxes = _native.argmax_rows(logits.detach()).transpose(0, 1).cpu().long()
        decoded = self.process_sequences(maxes, encoded_lens)
        ret = {'decoded': decoded, 'decoded_frames': maxes, 'logits': logits}
        if ctc_loss is not None:
            ret['loss'] = {'ctc_loss': ctc_loss, 'loss': ctc_loss}
        if return_texts_and_generated_loss:
            decoded_lens = torch.IntTensor([len(x) for x in decoded])
            ctc_generated_losses = self.get_ctc_losses(
                logits, encoded_lens, decoded, decoded_lens, {})
            ret['text_loss'] = ctc_text_losses.tolist()
            ret['generated_loss'] = ctc_generated_losses.tolist()
        if return_logits_text_diff:
            ret['logits_text_diff'] = (torch.as_tensor(encoded_lens) -
                                       torch.as_tensor(text_lens)).tolist()
        return ret

    def process_sequences(self, logits, logits_lens):
        return [self.process_sequence(logits[i, :], logits_lens[i])
                for i in range(len(logits_lens))]

    def process_sequence(self, logits, logits_len):
        """Bug-compatible with the reference's default branch (:385-391): the
        `or` makes the adjacent-frame test vacuous for i != 0; for i == 0 the
        frame is compared with the LAST element of the padded row; a symbol is
        dropped when it equals (mod num_symbols) the previously kept one."""
        if self.fix_greedy_decoder:
            # the reference calls an undefined remove_repetitions_blanks (:381)
            raise NameError("name 'remove_repetitions_blanks' is not defined")
        frames = [int(c) for c in logits.tolist()]
        blanks = set(self.blanks)
        ret = []
        for i in range(int(logits_len)):
            char = frames[i]
            if char not in blanks and (i != 0 or char != frames[i - 1]):
                if not ret or (ret[-1] % self.num_symbols != char % self.num_symbols):
                    ret.append(char)
        return ret


class _SubRowMax(torch.autograd.Function):
    """logits - max_c(logits).detach() and sum_t max*mask (reference :479-484).
    The maximum is detached in the reference, so the gradient passes through."""

    @staticmethod
    def forward(ctx, logits, lens_dev):
        y, _, max_sum = _native.sub_rowmax(logits.contiguous(), lens_dev)
        ctx.mark_non_differentiable(max_sum)
        return y, max_sum

    @staticmethod
    def backward(ctx, dy, _dsum):
        return dy, None


class FSTDecoder(BaseDecoder):
    """reference :394-593"""

    def __init__(self, sample_batch, num_classes,
                 graph_generator, normalize_by_dim=None,
                 numerator_red='logsumexp', denominator_red='logsumexp',
                 embedder='LutLinear', embedder_kwargs={}, **kwargs):
        super(FSTDecoder, self).__init__(**kwargs)
        self.graph_generator = utils.contruct_from_kwargs(
            graph_generator, 'att_speech.fst_utils',
            {'num_classes': num_classes, 'num_symbols': self.num_symbols})
        self.context_order = self.graph_generator.context_order
        self.normalize_by_dim = normalize_by_dim
        self.numerator_red = numerator_red
        self.denominator_red = denominator_red
        self._verify = False
        self.verbose = False
        if self.normalize_by_dim not in [None, 0]:
            assert (self.graph_generator.num_classes ==
                    self.graph_generator.num_symbols ** self.context_order)
        if self.num_symbols is None:
            self.num_symbols = self.graph_generator.num_symbols
        # transducer used to read the labels off the best state sequence
        self.dec_fst = self.graph_generator.decoding_fst
        rnn_hidden_size = sample_batch["features"].size()[2]
        embedder = globals()[embedder]
        ngram_to_class = self.graph_generator.ngram_to_class
        modules = [embedder(rnn_hidden_size, self.graph_generator.num_symbols,
                            ngram_to_class, **embedder_kwargs)]
        fully_connected = nn.Sequential(*modules)
        self.fc = nn.Sequential(SequenceWise(fully_connected))

    def logits(self, encoded, encoded_lens=None, extra_ret=None):
        logits = self.fc(encoded)
        if extra_ret is not None:
            extra_ret['unnormed_logits'] = logits
        if self.normalize_by_dim is not None:
            logits = get_normalized_acts(
                logits, encoded_lens, self.num_symbols, self.context_order,
                self.normalize_by_dim, normalize_logits=True)
        return logits

    def get_fst_loss(self, logits, encoded_lens, texts, text_lens,
                     other_data_in_batch):
        if other_data_in_batch and 'graph_matrices' in other_data_in_batch:
            numerator_matrices = other_data_in_batch['graph_matrices']
            if not self._verify:                                   # (:460-468)
                numerator_matrices2 = \
                    self.graph_generator.get_training_matrices_batch(
                        texts, text_lens, 'cpu')
                assert max([torch.abs(m1.cpu() - m2).max().item()
                            for m1, m2 in zip(numerator_matrices,
                                              numerator_matrices2)]) < 1e-10
                self._verify = True
        else:
            # no graphs in the batch: build the lattices on the device from the
            # labels (the reference builds them on the host here, :470-471)
            numerator_matrices = self.graph_generator.get_training_graph_device(
                texts, text_lens, logits.device)
        lens_dev = torch.as_tensor(encoded_lens).to(logits.device, torch.int32)
        logits, logits_sum = _SubRowMax.apply(logits, lens_dev)     # (:479-484)
        neg_inf = self.graph_generator.nc_weight
        numerator_loss = -fst_utils.path_reduction(
            logits, encoded_lens, numerator_matrices,
            red_kind=self.numerator_red, neg_inf=neg_inf)
        if self.denominator_red != 'none':
            denominator_matrices = self.graph_generator.get_decoding_matrices('cpu')
            denominator_loss = -fst_utils.path_reduction(
                logits, encoded_lens, denominator_matrices,
                red_kind=self.denominator_red, neg_inf=neg_inf)
        else:
            denominator_loss = logits_sum
        if self.verbose:
            print("global loss: [loss: num %g, den %g, com %g]" % (
                numerator_loss.sum().item(), denominator_loss.sum().item(),
                -logits_sum.sum().item()))
        return numerator_loss - denominator_loss

    def forward(self, encoded, encoded_lens, texts, text_lens, spkids=None,
                **other_data_in_batch):
        extra_ret = {}
        logits = self.logits(encoded, encoded_lens, extra_ret=extra_ret)
        fst_losses = self.get_fst_loss(
            logits, encoded_lens, texts, text_lens, other_data_in_batch)
        fst_loss = fst_losses.sum()
        return {'fst_loss': fst_loss, 'loss': fst_loss}

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_texts_and_generated_loss=False,
               return_logits_text_diff=False, spkids=None,
               **other_data_in_batch):
        logits = self.logits(encoded, encoded_lens)
        denominator_matrices = self.graph_generator.get_decoding_matrices('cpu')
        # best state sequence of the decoding graph; the reference gets the
        # same indices from the autograd gradient of the Viterbi score (:546-554)
        _, selidx = fst_utils.viterbi_path(
            logits.detach(), encoded_lens, denominator_matrices,
            self.graph_generator.nc_weight)
        selidx = selidx.cpu().numpy()
        lens = [int(l) for l in torch.as_tensor(encoded_lens).tolist()]
        decoded_texts = [self.dec_fst.read_out(selidx[:lens[i], i])
                         for i in range(logits.size(1))]          # (:556-571)
        ret = {'decoded': decoded_texts, 'logits': logits}
        if texts is not None and text_lens is not None:
            fst_text_losses = self.get_fst_loss(
                logits, encoded_lens, texts, text_lens, other_data_in_batch)
            fst_text_loss = fst_text_losses.sum()
            ret['loss'] = dict(fst_loss=fst_text_loss, loss=fst_text_loss)
        if return_texts_and_generated_loss:
            decoded_lens = torch.IntTensor([len(x) for x in decoded_texts])
            fst_generated_losses = self.get_fst_loss(
                logits, encoded_lens, decoded_texts, decoded_lens, None)
            ret['text_loss'] = fst_text_losses.tolist()
            ret['generated_loss'] = fst_generated_losses.tolist()
        if return_logits_text_diff:
            ret['logits_text_diff'] = (torch.as_tensor(encoded_lens) -
                                       torch.as_tensor(text_lens)).tolist()
        return ret
