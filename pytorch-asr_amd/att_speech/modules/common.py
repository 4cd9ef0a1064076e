from torch import nn


class SequenceWise(nn.Module):
    """Collapses T*BS*F to (T*BS)*F and applies a module
    (reference modules/common.py:4-25)."""

    def __init__(self, module):
        super(SequenceWise, self).__init__()
        self.module = module

    def forward(self, x):
        time, batch_size = x.size(0), x.size(1)
        x = x.reshape(time * batch_size, -1)
        x = self.module(x)
        x = x.view(time, batch_size, -1)
        return x
