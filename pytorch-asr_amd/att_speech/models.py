"""reference att_speech/models.py:21-97 — SpeechModel = encoder o decoder."""
from __future__ import absolute_import, division, print_function

import itertools

import torch
from torch import nn

from att_speech import utils


class SpeechModel(nn.Module):
    def __init__(self, encoder, decoder, sample_batch, num_classes, vocabulary,
                 **kwargs):
        super(SpeechModel, self).__init__(**kwargs)
        self.encoder = utils.contruct_from_kwargs(
            encoder, 'att_speech.modules.encoders', {'sample_batch': sample_batch})
        with torch.no_grad():                       # size probe (:29-30)
            was_training = self.encoder.training
            self.encoder.eval()
            sample_batch["features"], sample_batch["features_lens"] = (
                self.encoder(**_encoder_kwargs(sample_batch)))
            self.encoder.train(was_training)
        self.decoder = utils.contruct_from_kwargs(
            decoder, 'att_speech.modules.decoders',
            {'sample_batch': sample_batch, 'num_classes': num_classes,
             'vocabulary': vocabulary})

    def load_state(self, state_dict, strict=True):
        self.load_state_dict(state_dict, strict)

    def forward(self, features, feature_lens, spkids, texts, text_lens,
                ivectors=None, **kwargs):
        dec_kwargs = kwargs
        gg = getattr(self.decoder, 'graph_generator', None)
        if (features.is_cuda and self.training and texts is not None and 'graph_matrices' not in kwargs
                and hasattr(gg, 'get_training_graph_device')
                and hasattr(self.decoder, '_numerator_graphs')):
            # the numerator lattices are built on the device from the labels; doing it here, at
            # the start of the step, keeps the labels' host-to-device copy (which waits for the
            # queue to drain) out of the decoder, where it would sit behind the whole encoder
            dec_kwargs = dict(kwargs, graph_matrices=gg.get_training_graph_device(
                texts, text_lens, features.device))
        encoded, encoded_lens = self.encoder(features, feature_lens, spkids,
                                             ivectors, **kwargs)
        return self.decoder(encoded, encoded_lens, texts, text_lens,
                            spkids=spkids, **dec_kwargs)

    def decode(self, features, feature_lens, speakers, texts=None,
               text_lens=None, encoder_args=None, decoder_args=None,
               ivectors=None, **kwargs):
        """(:63-93) extra keyword arguments go to BOTH halves, on top of the
        per-half dictionaries; runs without autograd."""
        enc_kw = dict(encoder_args or {}, **kwargs)
        dec_kw = dict(decoder_args or {}, **kwargs)
        with torch.no_grad():
            enc, enc_lens = self.encoder(features, feature_lens, speakers, ivectors, **enc_kw)
            return self.decoder.decode(enc, enc_lens, texts, text_lens, spkids=speakers, **dec_kw)

    def get_parameters_for_optimizer(self):
        return itertools.chain(self.encoder.get_parameters_for_optimizer(),
                               self.decoder.parameters())


def _encoder_kwargs(sample_batch):
    """the reference calls self.encoder(**sample_batch) with keys
    features / features_lengths / spkids (models.py:29-30)."""
    kw = dict(sample_batch)
    if 'features_lengths' not in kw and 'features_lens' in kw:
        kw['features_lengths'] = kw.pop('features_lens')
    kw.setdefault('spkids', None)
    return kw
