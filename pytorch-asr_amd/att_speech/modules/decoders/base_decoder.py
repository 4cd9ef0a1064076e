"""Decoder interface (reference modules/decoders/base_decoder.py:18-41): a decoder
knows its symbol inventory (`vocabulary` + the implicit '<eos>', `num_symbols` is the
inventory size without it, or None when no vocabulary was given) and implements
`forward` (training loss dict) and `decode`."""
from torch import nn


class BaseDecoder(nn.Module):
    def __init__(self, vocabulary=None, **kwargs):
        super(BaseDecoder, self).__init__(**kwargs)
        symbols = [] if vocabulary is None else list(vocabulary)
        self.num_symbols = None if vocabulary is None else len(symbols)
        self.vocabulary = symbols + ['<eos>']

    def forward(self, encoded, encoded_lens, texts, text_lens, **kwargs):
        raise NotImplementedError("%s.forward" % type(self).__name__)

    def decode(self, encoded, encoded_lens, **kwargs):
        raise NotImplementedError("%s.decode" % type(self).__name__)
