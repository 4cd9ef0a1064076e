from torch import nn


class BaseEncoder(nn.Module):
    """reference modules/encoders/base_encoder.py"""

    def __init__(self, **kwargs):
        if len(kwargs) > 0:
            raise RuntimeError(
                "Unrecognized options: {}".format(', '.join(kwargs.keys())))
        super(BaseEncoder, self).__init__()

    def forward(self, features, features_lengths, spkids):
        raise NotImplementedError

    def get_parameters_for_optimizer(self):
        return self.parameters()
