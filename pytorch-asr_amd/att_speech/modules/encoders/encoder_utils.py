"""reference modules/encoders/encoder_utils.py: Normalization (:16-52),
BatchRNN (:55-124), SequentialWithOptionalAttributes (:127-133).

The recurrent layers work on PADDED [T,B,F] tensors plus lengths instead of
PackedSequence objects: padding frames are masked inside the recurrence, which
yields the same values on every valid frame (a packed bidirectional LSTM starts
its reverse direction at each utterance's own last frame — so does the masked
one) and lets the input projection run as one dense [T*B, F] GEMM."""
from __future__ import division, print_function

import inspect
import os
import warnings

import torch
from torch import nn


class Identity(nn.Module):
    def __init__(self, *args, **kwargs):
        super(Identity, self).__init__()

    def forward(self, x):
        return x


_NORM_LAYERS = {('batch_norm', 1): nn.BatchNorm1d, ('batch_norm', 2): nn.BatchNorm2d,
                ('instance_norm', 1): nn.InstanceNorm1d, ('instance_norm', 2): nn.InstanceNorm2d}


class Normalization(nn.Module):
    """`norm_type` in {batch_norm, instance_norm, none / empty} over `nary`-dimensional
    feature maps; the layer is the attribute `batch_norm` whatever its kind (state_dict
    prefix `...batch_norm.*`, reference :16-52)."""

    def __init__(self, norm_type, nary, input_size):
        super(Normalization, self).__init__()
        self.nary = nary
        kind = norm_type or 'none'
        if kind == 'none':
            self.batch_norm = Identity()
        elif kind not in ('batch_norm', 'instance_norm'):
            raise ValueError("Unknown normalization type {}. Possible are: batch_norm, "
                             "instance_norm or none".format(norm_type))
        elif (kind, nary) not in _NORM_LAYERS:
            raise ValueError("Unknown nary for {} normalization".format(norm_type))
        else:
            self.batch_norm = _NORM_LAYERS[kind, nary](input_size)

    def forward(self, x, speaker=None):
        return self.batch_norm(x)


class BatchRNN(nn.Module):
    """One (bi)directional recurrent layer, bias-free, directions summed
    (reference encoder_utils.py:55-124).  `self.rnn` is an nn.LSTM / nn.GRU so
    the state_dict keys (rnn.weight_ih_l0, rnn.weight_hh_l0, *_reverse) are the
    reference's; input is (padded [T,B,F], lens [B])."""

    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM,
                 bidirectional=False, packed_data=False, normalization=None,
                 projection_size=0, residual=False, subsample=False):
        super(BatchRNN, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.residual = residual
        self.batch_norm = Normalization(normalization, 1, input_size)
        self.rnn = rnn_type(input_size=input_size, hidden_size=hidden_size,
                            bidirectional=bidirectional, bias=False)
        self.num_directions = 2 if bidirectional else 1
        self.subsample = subsample
        if projection_size > 0:
            self.projection = torch.nn.Linear(
                hidden_size * self.num_directions, projection_size, bias=False)
        else:
            self.projection = None

    def flatten_parameters(self):
        self.rnn.flatten_parameters()

    def _use_native(self, x):
        # the MI355X path: hand-written recurrence kernels (csrc/lstm.hip)
        native = (isinstance(self.rnn, nn.LSTM) and self.bidirectional
                  and self.hidden_size in (64, 128, 256, 320, 384, 512, 768)
                  and not self.rnn.bias)
        if x.is_cuda and not native and not getattr(self, '_warned', False):
            self._warned = True
            warnings.warn(
                'BatchRNN(%s, hidden %d, bidirectional=%s): no hand-written recurrence for this '
                'layer (built: bias-free bidirectional LSTM, hidden 64/128/256/320/384/512/768); '
                'running torch nn.%s (MIOpen) instead' % (
                    type(self.rnn).__name__, self.hidden_size, self.bidirectional,
                    type(self.rnn).__name__))
        return x.is_cuda and native

    def forward(self, x, lens, speakers=None):
        """x [T,B,F] padded, lens [B] (CPU int, sorted descending)."""
        T, B, _ = x.shape
        lens_t = torch.as_tensor(lens)
        if self.residual:
            res = x
        if not isinstance(self.batch_norm.batch_norm, Identity):
            # the reference normalises the packed data, i.e. valid frames only
            mask = (torch.arange(T)[:, None] < lens_t[None, :]).to(x.device)
            flat = x[mask]
            x = x.clone()
            x[mask] = self.batch_norm(flat)
        summed = False
        if self._use_native(x):
            from att_speech.modules.encoders.native_lstm import bilstm
            # the direction sum of :112-117 happens inside the function when nothing
            # sits between the LSTM and the merge
            summed = self.projection is None and not self.subsample
            y = bilstm(x, lens_t, self.rnn, sum_dirs=summed)
            y = y if summed else y.view(T, B, -1)               # [T,B,2H], zeros on padding
        else:
            # host / non-LSTM evaluation with stock torch ops (CPU reference in
            # the tests and bench.py's cpu_baseline; GRU layers)
            packed = nn.utils.rnn.pack_padded_sequence(x, lens_t.cpu())
            y, _ = self.rnn(packed)
            y, _ = nn.utils.rnn.pad_packed_sequence(y, total_length=T)
        if self.subsample:
            y = y[::2]
            lens_t = lens_t // 2
        if self.projection is not None:
            y = self.projection(y)
        elif self.bidirectional and not summed:
            y = y.view(y.size(0), y.size(1), 2, -1).sum(2)      # (T,B,2H) -> (T,B,H)
        if self.residual:
            y = torch.nn.functional.relu(y + res)
        return y, lens_t


class SequentialWithOptionalAttributes(nn.Sequential):
    """reference :127-133 (py3: inspect instead of func_code); modules that
    take (x, lens, ...) return (x, lens)."""

    def _plain_native_stack(self, x):
        """every module a BatchRNN that is nothing but a native bidirectional LSTM with summed
        directions: the whole stack runs as one autograd node (native_lstm.bilstm_stack)"""
        mods = list(self._modules.values())
        return len(mods) > 1 and all(
            isinstance(m, BatchRNN) and m._use_native(x) and m.projection is None
            and not m.subsample and not m.residual
            and isinstance(m.batch_norm.batch_norm, Identity) for m in mods)

    def forward(self, input, lens, *args):
        if os.environ.get('ASR_LSTM_STACK', '1') != '0' and self._plain_native_stack(input):
            from att_speech.modules.encoders.native_lstm import bilstm_stack
            lens_t = torch.as_tensor(lens)
            return bilstm_stack(input, lens_t, [m.rnn for m in self._modules.values()]), lens_t
        for module in self._modules.values():
            if isinstance(module, BatchRNN):
                nparams = len(inspect.signature(module.forward).parameters)
                input, lens = module(input, lens, *args[:max(0, nparams - 2)])
            else:
                input = module(input)
        return input, lens
