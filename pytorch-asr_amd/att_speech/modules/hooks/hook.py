"""Training-loop hook interface (reference modules/hooks/hook.py:1-24).

A hook is an object with a `priority` and six optional call-backs; the step driver
(`att_speech.dp.train_step`, reference trainer.py:229-272) invokes them by keyword:

    pre_run(model, optimizer)                                  once, before training
    pre_train_forward(model, optimizer, current_iteration)     before every forward
    pre_backward(model, optimizer, current_iteration, loss)    truthy result: skip the step
    post_backward(model, optimizer, current_iteration, loss)   sees the (all-reduced) gradients;
                                                               truthy result: skip the step
    post_optimizer_step(model, optimizer, current_iteration, loss)
    post_dev_eval(model, current_iteration, logger, save_dir, dev_dataset)

Every call-back defaults to a no-op returning None."""

_CALLBACKS = ('pre_run', 'pre_train_forward', 'pre_backward', 'post_backward',
              'post_optimizer_step', 'post_dev_eval')


def _noop(self, *args, **kwargs):
    return None


class TrainingLoopHook(object):
    def __init__(self, priority=None):
        self.priority = priority or 0


for _name in _CALLBACKS:
    setattr(TrainingLoopHook, _name, _noop)
del _name
