"""reference modules/common.py:4-25 — SequenceWise: run a frame-wise module over a
`[T, B, F]` sequence by flattening the two leading dimensions.  The wrapped module
is the attribute `module` (state_dict prefix `...module.*`)."""
from torch import nn


class SequenceWise(nn.Module):
    def __init__(self, module):
        super(SequenceWise, self).__init__()
        self.module = module

    def forward(self, x):
        steps, batch = x.shape[:2]
        y = self.module(x.reshape(steps * batch, -1))
        return y.view(steps, batch, -1)
