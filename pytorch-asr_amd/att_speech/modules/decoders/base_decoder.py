from torch import nn


class BaseDecoder(nn.Module):
    """reference modules/decoders/base_decoder.py:18-41"""

    def __init__(self, vocabulary=None, **kwargs):
        super(BaseDecoder, self).__init__(**kwargs)
        if vocabulary is None:
            vocabulary = []
            self.num_symbols = None
        else:
            self.num_symbols = len(vocabulary)
        self.vocabulary = list(vocabulary) + ['<eos>']

    def forward(self, encoded, encoded_lens, texts, text_lens, **kwargs):
        raise NotImplementedError

    def decode(self, encoded, encoded_lens, **kwargs):
        raise NotImplementedError
