"""Training-step hooks: GradientClipping incl. skip-step and
PolyakDecay against their definitions in the reference
(modules/hooks/gradient_clipping.py:13-53, polyak.py:6-59), and the step order
of trainer.py:229-272 in att_speech.dp.train_step."""
import copy

import numpy as np
import torch
from torch import nn


class Toy(nn.Module):
    def __init__(self):
        super(Toy, self).__init__()
        torch.manual_seed(0)
        self.net = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.Tanh(), nn.Linear(5, 3))

    def forward(self, x):
        return {'loss': (self.net(x) ** 2).sum()}

    def get_parameters_for_optimizer(self):
        return self.parameters()


def _grads(model, x):
    model.zero_grad()
    model(x)['loss'].backward()
    return [p.grad.clone() for p in model.parameters()]


def test_gradient_clipping_matches_clip_grad_norm_and_skips():
    from att_speech.dp import FlatGradBucket
    from att_speech.modules.hooks import GradientClipping
    x = torch.randn(8, 6)
    ref = Toy()
    g = _grads(ref, x)
    total = float(torch.sqrt(sum((t ** 2).sum() for t in g)))
    for use_bucket in (False, True):
        for clip, skip_at, want_skip in [(total * 0.5, np.inf, False), (total * 2, np.inf, False),
                                         (total * 0.5, total * 0.9, True)]:
            m = Toy()
            hook = GradientClipping(clip, skip_at)
            if use_bucket:
                hook.bucket = FlatGradBucket(m.parameters())
                hook.bucket.zero_()
            m(x)['loss'].backward()
            skipped = hook.post_backward(model=m, optimizer=None, current_iteration=1, loss=None)
            assert skipped is want_skip
            coef = min(1.0, clip / (total + 1e-6))
            for p, t in zip(m.parameters(), g):
                torch.testing.assert_close(p.grad, t * coef, rtol=1e-5, atol=1e-7)
            assert len(hook.gstats.norms) == 1 and abs(hook.gstats.norms[0] - total) < 1e-4 * total
            assert hook.gstats.clips == int(total > clip) and hook.gstats.skips == int(want_skip)


def test_polyak_decay_is_the_reference_recurrence():
    from att_speech.modules.hooks import PolyakDecay
    m = Toy()
    hook = PolyakDecay([0.9, 0.5])
    hook.pre_run(m, None)
    want = {d: copy.deepcopy(m.state_dict()) for d in (0.9, 0.5)}
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for it in range(3):
        opt.zero_grad()
        m(torch.randn(8, 6))['loss'].backward()
        opt.step()
        hook.post_optimizer_step(m, opt, it, None)
        st = m.state_dict()
        for d in want:
            for k in want[d]:
                if want[d][k].is_floating_point():
                    want[d][k] = d * want[d][k] + (1 - d) * st[k]      # polyak.py:31-37
                else:
                    want[d][k] = st[k].clone()
    for d in want:
        avg = getattr(m, 'avg_state_dict_%f' % d)
        assert set(avg) == set(m.state_dict())
        for k in avg:
            torch.testing.assert_close(avg[k], want[d][k], rtol=1e-5, atol=1e-7)
    # a rate that is no longer configured is dropped on the next run (polyak.py:20-24)
    PolyakDecay([0.9]).pre_run(m, None)
    assert hasattr(m, 'avg_state_dict_%f' % 0.9) and not hasattr(m, 'avg_state_dict_%f' % 0.5)


def test_train_step_order_and_skip():
    from att_speech.dp import FlatGradBucket, train_step
    from att_speech.modules.hooks import GradientClipping, TrainingLoopHook
    calls = []

    class Spy(TrainingLoopHook):
        def pre_train_forward(self, **kw):
            calls.append('pre_fwd')

        def pre_backward(self, **kw):
            calls.append('pre_bwd')

        def post_backward(self, model, **kw):
            calls.append('post_bwd:%s' % all(p.grad is not None for p in model.parameters()))

        def post_optimizer_step(self, **kw):
            calls.append('post_step')

    m = Toy()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    bucket = FlatGradBucket(m.parameters())
    x = torch.randn(8, 6)
    before = [p.detach().clone() for p in m.parameters()]
    _, skipped = train_step(m, opt, ((x,), {}), hooks=[Spy(), GradientClipping(1e-3, 1e-6)],
                            bucket=bucket, current_iteration=1)
    assert skipped and calls == ['pre_fwd', 'pre_bwd', 'post_bwd:True', 'post_step']
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)                       # optimizer step skipped
    _, skipped = train_step(m, opt, ((x,), {}), hooks=[GradientClipping(1e-3)], bucket=bucket)
    assert not skipped and abs(float(bucket.flat.norm()) - 1e-3) < 1e-6
    assert any(not torch.equal(p, b) for p, b in zip(m.parameters(), before))


def test_gradient_clipping_skips_a_non_finite_norm():
    """`nan > skip_step_norm` is False: the reference would take the optimizer step with NaN
    gradients.  With data parallelism one rank's NaN (a timed-out LSTM hand-off) is in every
    rank's bucket after the all-reduce, so a non-finite norm always skips the step."""
    from att_speech.modules.hooks import GradientClipping

    class M(torch.nn.Linear):
        def get_parameters_for_optimizer(self):
            return self.parameters()
    m = M(3, 2)
    for p in m.parameters():
        p.grad = torch.full_like(p, float('nan'))
    hook = GradientClipping(clip_norm=1.0, skip_step_norm=10.0)
    assert hook.post_backward(model=m, optimizer=None, current_iteration=0, loss=None) is True
