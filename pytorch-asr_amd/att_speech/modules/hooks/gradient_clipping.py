"""reference modules/hooks/gradient_clipping.py:13-53 — clip the global gradient
norm to `clip_norm`; ask the trainer to skip the optimizer step when the
unclipped norm exceeds `skip_step_norm`; running min/mean/max + clip/skip
fractions go to the logger.

Data parallel: the hook runs in `post_backward`, i.e. AFTER the gradient
all-reduce of `att_speech.dp.train_step`, so the norm is the global one on every
rank and all ranks take the same clip / skip decision.  When the gradients are
views into a `FlatGradBucket` the norm and the scaling are one kernel each over
the flat buffer instead of one per parameter."""
import numpy as np
import torch
from torch.nn.utils import clip_grad_norm_

from att_speech.logger import DefaultTensorLogger
from att_speech.modules.hooks.hook import TrainingLoopHook

logger = DefaultTensorLogger()


class GradientClipping(TrainingLoopHook):
    def __init__(self, clip_norm, skip_step_norm=np.inf, **kwargs):
        self.clip_norm = clip_norm
        self.skip_step_norm = skip_step_norm
        self.gstats = None
        self.bucket = None          # set by dp.train_step when a flat bucket is in use
        super(GradientClipping, self).__init__(**kwargs)

    def _clip(self, model):
        b = self.bucket
        if b is not None:
            b.check_views()
            norm = float(b.flat.norm(2))
            # same rule as clip_grad_norm_: scale by clip/(norm + 1e-6) when that is < 1
            coef = self.clip_norm / (norm + 1e-6)
            if coef < 1:
                b.flat.mul_(coef)
            return norm
        return float(clip_grad_norm_(model.get_parameters_for_optimizer(), self.clip_norm))

    def post_backward(self, model, optimizer, current_iteration, loss):
        unclipped_norm = self._clip(model)
        clipped = int(unclipped_norm > self.clip_norm)
        skipped = int(unclipped_norm > self.skip_step_norm)
        if self.gstats is None:
            self.gstats = (1, unclipped_norm, unclipped_norm, unclipped_norm, clipped, skipped)
        else:
            n, g_min, g_sum, g_max, n_clip, n_skip = self.gstats
            self.gstats = (n + 1, min(g_min, unclipped_norm), g_sum + unclipped_norm,
                           max(g_max, unclipped_norm), n_clip + clipped, n_skip + skipped)
        if logger.is_currently_logging():
            n, g_min, g_sum, g_max, n_clip, n_skip = self.gstats
            logger.log_scalar("gclip/min", g_min)
            logger.log_scalar("gclip/max", g_max)
            logger.log_scalar("gclip/mean", 1.0 * g_sum / n)
            logger.log_scalar("gclip/clipfrac", 1.0 * n_clip / n)
            logger.log_scalar("gclip/skipfrac", 1.0 * n_skip / n)
            self.gstats = None
        if clipped:
            print("Grad clipped by ", 1.0 * self.clip_norm / unclipped_norm)
        return bool(skipped)            # tells the trainer to skip this step
