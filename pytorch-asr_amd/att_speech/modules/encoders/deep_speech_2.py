"""reference modules/encoders/deep_speech_2.py:14-160 — conv2d+BN+Hardtanh x2
then bidirectional recurrent layers with summed directions."""
import warnings
from collections import OrderedDict

import torch
from torch import nn

from att_speech.modules.encoders.base_encoder import BaseEncoder
from att_speech.modules.encoders.encoder_utils import (
    BatchRNN, Normalization, SequentialWithOptionalAttributes)


class DeepSpeech2(BaseEncoder):
    def __init__(self, sample_batch, conv_normalization='batch_norm',
                 conv_strides=[[2, 2], [2, 1]],
                 conv_kernel_sizes=[[41, 11], [21, 11]],
                 conv_num_features=[32, 32],
                 rnn_hidden_size=768, rnn_nb_layers=5, rnn_projection_size=0,
                 rnn_type=nn.LSTM, rnn_dropout=0.0, rnn_residual=False,
                 rnn_normalization='batch_norm', rnn_subsample=None, **kwargs):
        super(DeepSpeech2, self).__init__(**kwargs)
        import os
        self.conv_bf16 = os.environ.get('ASR_CONV_BF16', '1') != '0'
        self.fused_bn = os.environ.get('ASR_FUSED_BN', '1') != '0'
        self.native_conv = os.environ.get('ASR_NATIVE_CONV', '1') != '0'
        if isinstance(rnn_type, str):
            rnn_type = {'LSTM': nn.LSTM, 'GRU': nn.GRU}[rnn_type.split('.')[-1]]
        self.makeConv(sample_batch, conv_strides, conv_kernel_sizes,
                      conv_num_features, conv_normalization)
        features = sample_batch['features']
        features = features.permute(0, 3, 1, 2)
        with torch.no_grad():
            was_training = self.conv.training
            self.conv.eval()        # size probe only: do not touch BN statistics
            after_conv_size = self.conv.forward(features).size()
            self.conv.train(was_training)
        self.rnn_input_size = after_conv_size[1] * after_conv_size[3]
        self.makeRnn(rnn_hidden_size, rnn_nb_layers, rnn_projection_size,
                     rnn_type, rnn_dropout, rnn_residual, rnn_normalization,
                     rnn_subsample)

    def makeConv(self, sample_batch, conv_strides, conv_kernel_sizes,
                 conv_num_features, normalization):
        num_channels = sample_batch['features'].size()[3]
        conv_padding = int(0.5 * (
            conv_kernel_sizes[1][0] - 1 +
            conv_strides[0][0] * (conv_kernel_sizes[0][0] - 1)))      # (:56-58)
        self.conv_cumative_stride = conv_strides[0][0] * conv_strides[1][0]
        self.conv = nn.Sequential(
            nn.Conv2d(num_channels, conv_num_features[0],
                      kernel_size=tuple(conv_kernel_sizes[0]),
                      stride=tuple(conv_strides[0]), padding=(conv_padding, 0)),
            Normalization(normalization, 2, conv_num_features[0]),
            nn.Hardtanh(0, 20, inplace=True),
            nn.Conv2d(conv_num_features[0], conv_num_features[1],
                      kernel_size=tuple(conv_kernel_sizes[1]),
                      stride=tuple(conv_strides[1])),
            Normalization(normalization, 2, conv_num_features[1]),
            nn.Hardtanh(0, 20, inplace=True))

    def makeRnn(self, rnn_hidden_size, rnn_nb_layers, rnn_projection_size,
                rnn_type, rnn_dropout, rnn_residual, normalization,
                rnn_subsample):
        if rnn_subsample is None:
            rnn_subsample = []
        rnn_dropout = nn.Dropout(p=rnn_dropout) if rnn_dropout > 0.0 else None
        rnns = [('0', BatchRNN(input_size=self.rnn_input_size,
                               hidden_size=rnn_hidden_size, rnn_type=rnn_type,
                               bidirectional=True, packed_data=True,
                               normalization=None,
                               projection_size=rnn_projection_size,
                               subsample=(0 in rnn_subsample)))]
        for i in range(rnn_nb_layers - 1):
            rnn = BatchRNN(input_size=(rnn_hidden_size if rnn_projection_size == 0
                                       else rnn_projection_size),
                           projection_size=rnn_projection_size,
                           hidden_size=rnn_hidden_size, rnn_type=rnn_type,
                           bidirectional=True, packed_data=True,
                           normalization=normalization, residual=rnn_residual,
                           subsample=(i + 1 in rnn_subsample))
            if rnn_dropout:
                rnns.append(('{}_dropout'.format(i + 1), rnn_dropout))
            rnns.append(('{}'.format(i + 1), rnn))
        self.rnns = SequentialWithOptionalAttributes(OrderedDict(rnns))

    def _conv_forward(self, features):
        """conv stack on [B, ch, T, F] -> [T', B, C*F'] (the permute/view of
        reference :142-146 included).

        On the GPU (a) the convolutions take bf16 operands with fp32 accumulation,
        like the LSTM GEMMs (BASELINE config 2: bf16) — with the fused BatchNorm
        kernels both of them, their bf16 outputs being read directly; without, only
        the 32->32 channel one (97 % of the stack's flops);
        (b) BatchNorm2d + Hardtanh run as the fused kernels of csrc/bnact.hip: two
        passes over the convolution output forward and two backward, the clamped
        activation written straight as the next consumer's operand (bf16 NCHW for
        the second convolution, fp32 time-major for the LSTM stack)."""
        conv = self.conv
        fused = (features.is_cuda and self.fused_bn and len(conv) == 6
                 and isinstance(conv[0], nn.Conv2d) and isinstance(conv[3], nn.Conv2d)
                 and all(isinstance(conv[i], Normalization)
                         and isinstance(conv[i].batch_norm, nn.BatchNorm2d)
                         and conv[i].batch_norm.affine for i in (1, 4))
                 and all(isinstance(conv[i], nn.Hardtanh) for i in (2, 5)))
        bf16 = features.is_cuda and self.conv_bf16 and len(conv) == 6 \
            and isinstance(conv[3], nn.Conv2d)
        c2 = conv[3] if len(conv) == 6 else None

        def run_conv(c, x, with_bias=True, keep_bf16=False):
            bias = c.bias if with_bias else None
            if not bf16:
                return nn.functional.conv2d(x.float() if x.dtype != torch.float32 else x,
                                            c.weight, bias, c.stride, c.padding,
                                            c.dilation, c.groups)
            y = nn.functional.conv2d(
                x.to(torch.bfloat16), c.weight.to(torch.bfloat16),
                None if bias is None else bias.to(torch.bfloat16),
                c.stride, c.padding, c.dilation, c.groups)
            return y if keep_bf16 else y.float()

        def second_conv(x, with_bias=True):
            return run_conv(c2, x, with_bias)

        if fused:
            # the convolutions run bias-free; their biases are folded into the fused
            # BatchNorm kernels (no broadcast add, no full-tensor reduction for the
            # bias gradient)
            from att_speech.modules.encoders.native_bn import bn_hardtanh
            from att_speech.modules.encoders import native_conv
            c1 = conv[0]
            sums1 = sums2 = None                             # channel statistics from the conv epilogues
            if bf16 and self.native_conv and native_conv.first_supported(c1, features):
                x, sums1 = native_conv.conv1(features, c1)   # hand-written MFMA kernels
            else:
                if bf16 and self.native_conv and not getattr(self, '_warned_conv1', False):
                    self._warned_conv1 = True
                    warnings.warn('DeepSpeech2: the first convolution is not a (1 or 3)->32 7x7 stride-(1,2) '
                                  'shape on gradient-free features csrc/conv.hip is built for; using '
                                  'torch / MIOpen for it')
                x = run_conv(c1, features, with_bias=False, keep_bf16=True)
            x = bn_hardtanh(x, conv[1].batch_norm, conv[2], out_bf16=bf16, conv_bias=c1.bias,
                            chan_sums=sums1)
            if bf16 and self.native_conv and native_conv.supported(c2, x):
                y2, sums2 = native_conv.conv7x7c32(x, c2)
            else:
                if bf16 and self.native_conv and not getattr(self, '_warned_conv', False):
                    self._warned_conv = True
                    warnings.warn('DeepSpeech2: the second convolution is not the 32->32 7x7 '
                                  'stride-(3,1) shape csrc/conv.hip is built for; using torch / MIOpen')
                y2 = run_conv(c2, x, with_bias=False, keep_bf16=True)
            # the native LSTM rounds its input to bf16 first thing: hand it bf16 straight away
            # (half the bytes written here, no cast pass, bf16 gradient coming back)
            x = bn_hardtanh(y2, conv[4].batch_norm, conv[5], time_major=True, conv_bias=c2.bias,
                            chan_sums=sums2, out_bf16=bf16 and self._rnn_takes_bf16(y2))
            return x.view(x.size(0), x.size(1), -1)                  # [T', B, C*F']
        if bf16:
            x = conv[5](conv[4](second_conv(conv[2](conv[1](conv[0](features))))))
        else:
            x = conv(features)
        # bs x c x t x f -> t x bs x (c x f)
        x = x.permute(2, 0, 1, 3).contiguous()
        return x.view(x.size(0), x.size(1), -1)

    def _rnn_takes_bf16(self, x):
        """the first recurrent layer is the hand-written bf16-operand LSTM reading its input
        unmodified (ASR_LSTM_BF16_INPUT=0 keeps the fp32 hand-over)"""
        import os
        if os.environ.get('ASR_LSTM_BF16_INPUT', '1') == '0':
            return False
        first = next(iter(self.rnns._modules.values()), None)
        return (isinstance(first, BatchRNN) and first._use_native(x) and not first.residual
                and type(first.batch_norm.batch_norm).__name__ == 'Identity')

    def forward(self, features, features_lengths, spkids, ivectors=None,
                characteristic_vectors=None, **kwargs):
        # bs x t x f x c -> bs x c x t x f
        features = features.permute(0, 3, 1, 2)
        features_lengths = torch.as_tensor(features_lengths)
        features_lengths = ((features_lengths + self.conv_cumative_stride - 1)
                            // self.conv_cumative_stride).int()          # (:149-151)
        if features.is_cuda:
            # device copy of the lengths now, while the queue is short: the LSTM stack, the
            # class normalisation and the lattices all take it from here (_native.lens_on)
            from att_speech import _native
            _native.attach_device_lens(features_lengths, features.device)
        features = self._conv_forward(features)
        assert features_lengths[0] == features.size()[0]                 # (:152)
        return self.rnns(features, features_lengths, spkids)
