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


class _NormStats(object):
    """min / mean / max of the unclipped norm and the clip / skip fractions since
    the last time they were written to the logger."""

    def __init__(self):
        self.norms, self.clips, self.skips = [], 0, 0

    def add(self, norm, clipped, skipped):
        self.norms.append(norm)
        self.clips += clipped
        self.skips += skipped

    def write(self, log):
        n = float(len(self.norms))
        for name, value in (("min", min(self.norms)), ("max", max(self.norms)),
                            ("mean", sum(self.norms) / n), ("clipfrac", self.clips / n),
                            ("skipfrac", self.skips / n)):
            log.log_scalar("gclip/" + name, value)


class GradientClipping(TrainingLoopHook):
    def __init__(self, clip_norm, skip_step_norm=np.inf, **kwargs):
        super(GradientClipping, self).__init__(**kwargs)
        self.clip_norm, self.skip_step_norm = clip_norm, skip_step_norm
        self.gstats = None
        self.bucket = None          # set by dp.train_step when a flat bucket is in use

    def _clip(self, model):
        """Scale the gradient to norm <= clip_norm; returns the norm before scaling."""
        flat = None if self.bucket is None else self.bucket.flat
        if flat is None:
            return float(clip_grad_norm_(model.get_parameters_for_optimizer(), self.clip_norm))
        self.bucket.check_views()
        cached = getattr(self.bucket, 'cached_norm', None)      # dp.train_step's single read-back
        norm = float(flat.norm(2)) if cached is None else float(cached)
        scale = self.clip_norm / (norm + 1e-6)          # clip_grad_norm_'s rule
        if scale < 1:
            flat.mul_(scale)
        return norm

    def post_backward(self, model, optimizer, current_iteration, loss):
        norm = self._clip(model)
        too_large, skip = norm > self.clip_norm, norm > self.skip_step_norm
        if not np.isfinite(norm):
            # `nan > threshold` is False: the reference would step with NaN gradients; with a
            # shared bucket one rank's NaN is everybody's after the all-reduce: skip the step
            skip = True
        if self.gstats is None:
            self.gstats = _NormStats()
        self.gstats.add(norm, int(too_large), int(skip))
        if logger.is_currently_logging():
            self.gstats.write(logger)
            self.gstats = None
        if too_large:
            print("Grad clipped by ", self.clip_norm / norm)
        return bool(skip)               # truthy: the trainer skips this optimizer step
