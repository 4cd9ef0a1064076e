"""reference modules/hooks/polyak.py:6-59 — exponential moving averages of the
whole state_dict, one per decay rate, kept on the model as
`avg_state_dict_<decay %f>` (so the checkpointer saves them), updated after every
optimizer step; `post_dev_eval` scores the averaged models.

Here the update is one fused multi-tensor `lerp` per decay over all floating
tensors (avg += (1 - decay) * (cur - avg)) instead of two kernels per tensor;
integer buffers (BatchNorm's num_batches_tracked) just follow the live value."""
from copy import deepcopy

import torch

from att_speech.modules.hooks.hook import TrainingLoopHook


class PolyakDecay(TrainingLoopHook):
    def __init__(self, decay_rates, **kwargs):
        self.polyak_decay = decay_rates
        super(PolyakDecay, self).__init__(**kwargs)

    @staticmethod
    def dict_name(decay):
        return 'avg_state_dict_%f' % (decay,)

    def pre_run(self, model, optimizer):
        st_dict = model.state_dict()
        good = []
        for decay in self.polyak_decay:
            name = self.dict_name(decay)
            good.append(name)
            if not hasattr(model, name):
                setattr(model, name, deepcopy(st_dict))
        for name in list(model.__dict__.keys()):
            if name.startswith('avg_state_dict') and name not in good:
                print("Polyak deleting ", name)
                delattr(model, name)

    @torch.no_grad()
    def post_optimizer_step(self, model, optimizer, current_iteration, loss):
        st = model.state_dict()
        for decay in self.polyak_decay:
            avg = getattr(model, self.dict_name(decay))
            fa, fc = [], []
            for k, a in avg.items():
                if a.is_floating_point():
                    fa.append(a)
                    fc.append(st[k])
                else:
                    a.copy_(st[k])
            if fa:
                torch._foreach_lerp_(fa, fc, 1.0 - decay)

    def post_dev_eval(self, model, current_iteration, logger, save_dir, dev_dataset,
                      evaluate=None):
        """`evaluate(dev_dataset, model) -> dict` scores one model (the reference
        calls utils.evaluate_greedy, which belongs to the out-of-scope dev loop)."""
        if evaluate is None:
            return
        old_state = deepcopy(model.state_dict())
        for decay in self.polyak_decay:
            model.load_state_dict(getattr(model, self.dict_name(decay)))
            result = evaluate(dev_dataset, model)
            logger.make_step_log('{}/dev_polyak_{}/'.format(save_dir, decay), current_iteration)
            for k, v in result.items():
                logger.log_scalar('_' + k, v)
            logger.end_log()
        model.load_state_dict(old_state)
