"""att_speech.fused_step.FusedClipAdam (csrc/optim.hip) against what it stands for: the
reference's GradientClipping hook (modules/hooks/gradient_clipping.py:13-53: clip_grad_norm_,
skip above skip_step_norm) followed by torch.optim.Adam.step (trainer.py:262-266), evaluated on
the CPU with torch's own optimizer.  fp32 on both sides; the orders of the roundings differ
(fused multiply-adds on the device), hence 2e-6 relative
(plus an absolute term for moments that cancel to near zero)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(7,), (33, 65), (1024,), (1025,), (3, 700), (5000,), (2, 3, 4, 5)]


def _reference_step(params, opt, clip, skip_norm, err):
    """What trainer.py does around one optimizer step; returns (norm, clipped, skipped)."""
    norm = float(torch.nn.utils.clip_grad_norm_(params, clip))
    skipped = (not np.isfinite(norm)) or norm > skip_norm or err
    if not skipped:
        opt.step()
    return norm, norm > clip, skipped


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_fused_clip_adam_matches_hook_plus_torch_adam(wd):
    from att_speech.dp import FlatGradBucket
    from att_speech.fused_step import FusedClipAdam
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    cpu = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in SHAPES]
    gpu = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in cpu]
    opt = torch.optim.Adam(cpu, lr=3e-3, betas=(0.9, 0.98), eps=1e-7, weight_decay=wd)
    bucket = FlatGradBucket(gpu)
    clip, skip_norm = 50.0, 500.0
    fused = FusedClipAdam(bucket, lr=3e-3, betas=(0.9, 0.98), eps=1e-7, weight_decay=wd,
                          clip_norm=clip, skip_step_norm=skip_norm)
    err_word = torch.zeros(1, dtype=torch.int32, device=dev)
    # gradient scales: plain, clipped, plain, skipped (too large), NaN, LSTM error, plain, clipped
    plan = [(0.1, 0, 0), (3.0, 0, 0), (0.2, 0, 0), (100.0, 0, 0), (0.1, 1, 0), (0.1, 0, 1), (0.05, 0, 0), (1.0, 0, 0)]
    want = []
    for scale, nan, err in plan:
        grads = [torch.randn(*s, generator=g) * scale for s in SHAPES]
        if nan:
            grads[2][5] = float('nan')
        for p, q, gr in zip(cpu, gpu, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)                 # the views into the flat bucket
        err_word.fill_(err)
        fused.step(err_word)
        want.append(_reference_step(cpu, opt, clip, skip_norm, bool(err)))
    got = fused.drain()
    assert len(got) == len(plan)
    for (norm, clipped, skipped, err), (wn, wc, ws), (_, _, e) in zip(got, want, plan):
        if np.isfinite(wn):
            assert abs(norm - wn) <= 2e-6 * wn
            assert clipped == wc
        assert skipped == ws and err == bool(e)
    assert [r[2] for r in got] == [False, False, False, True, True, True, False, False]
    assert fused.steps_taken == 5
    for p, q in zip(cpu, gpu):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-6, atol=2e-7)
    # the moments, through the export into a torch optimizer's state
    opt2 = torch.optim.Adam(gpu, lr=3e-3)
    fused.export_state(opt2)
    for p, q in zip(cpu, gpu):
        np.testing.assert_allclose(opt2.state[q]['exp_avg'].cpu().numpy(), opt.state[p]['exp_avg'].numpy(),
                                   rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(opt2.state[q]['exp_avg_sq'].cpu().numpy(), opt.state[p]['exp_avg_sq'].numpy(),
                                   rtol=2e-6, atol=1e-8)
        assert int(opt2.state[q]['step']) == int(opt.state[p]['step']) == 5
    # ... and back (resuming): one more step from the imported state
    fused2 = FusedClipAdam(bucket, lr=3e-3, betas=(0.9, 0.98), eps=1e-7, weight_decay=wd,
                           clip_norm=clip, skip_step_norm=skip_norm)
    fused2.import_state(opt2)
    grads = [torch.randn(*s, generator=g) * 0.1 for s in SHAPES]
    for p, q, gr in zip(cpu, gpu, grads):
        p.grad = gr.clone()
        q.grad.copy_(gr)
    fused2.step(None)
    _reference_step(cpu, opt, clip, skip_norm, False)
    assert fused2.drain()[-1][2] is False and fused2.steps_taken == 6
    for p, q in zip(cpu, gpu):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_train_step_with_device_boundary_matches_host_boundary():
    """dp.train_step(fused=...) against the same step with the GradientClipping hook and
    torch.optim.Adam on the host side: same losses, same parameters after a few steps of a
    small SpeechModel (DeepSpeech2 + CTC lattice decoder), the hook's statistics fed late."""
    import copy
    from att_speech.dp import FlatGradBucket, train_step
    from att_speech.fused_step import FusedClipAdam
    from att_speech.modules.hooks import GradientClipping, PolyakDecay
    import bench
    dev = torch.device('cuda:0')
    B, T = 4, 700           # T' = 234 frames for transcripts of up to 100 labels
    feats, lens, texts, llens = bench.synthetic_batch(B, T, 0, 1)
    enc_cfg, dec_cfg = bench.model_config(1)
    from att_speech.models import SpeechModel
    torch.manual_seed(7)
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    m1 = SpeechModel(enc_cfg, dec_cfg, sb, 49, [str(i) for i in range(49)]).to(dev)
    m2 = copy.deepcopy(m1)
    p0 = [p.detach().float().cpu().numpy().copy() for p in m1.parameters()]
    fd = feats.to(dev)
    runs = []
    for model, device_side in ((m1, False), (m2, True)):
        bucket = FlatGradBucket(model.parameters())
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        hooks = [GradientClipping(clip_norm=30.0, skip_step_norm=1e6), PolyakDecay(decay_rates=[0.99])]
        for h in hooks:
            h.pre_run(model, opt)
        fused = FusedClipAdam.from_optimizer(opt, bucket, hooks[0]) if device_side else None
        losses = []
        for it in range(4):
            out, skip = train_step(model, opt, ((fd, lens, None, texts, llens), {}), hooks=hooks,
                                   bucket=bucket, fused=fused)
            assert not skip
            losses.append(float(out['loss']))
        if fused is not None:
            stats = fused.drain()
            assert len(stats) == 4 and not any(r[2] for r in stats) and any(r[1] for r in stats)
            assert hooks[0].gstats is not None and len(hooks[0].gstats.norms) == 4
        runs.append((losses, [p.detach().float().cpu().numpy() for p in model.parameters()],
                     getattr(model, PolyakDecay.dict_name(0.99))))
    (l1, p1, a1), (l2, p2, a2) = runs
    # same losses; the parameters agree as far as Adam allows: in its first steps an element
    # moves by ~lr per step whatever the size of its gradient (m / sqrt(v) = +-1), so elements
    # whose gradient is rounding noise (the two runs' backward passes differ by the order of
    # their atomic sums) move differently: compared as whole updates, not element by element
    np.testing.assert_allclose(l2, l1, rtol=1e-4)
    d1 = np.concatenate([(x - y).ravel() for x, y in zip(p1, p0)])
    d2 = np.concatenate([(x - y).ravel() for x, y in zip(p2, p0)])
    assert np.linalg.norm(d1 - d2) <= 0.03 * np.linalg.norm(d1)
    assert np.abs(d1 - d2).max() <= 4 * 1e-3          # nothing further apart than the four steps themselves
    a = np.concatenate([a1[k].float().cpu().numpy().ravel() for k in a1 if a1[k].is_floating_point()])
    b = np.concatenate([a2[k].float().cpu().numpy().ravel() for k in a2 if a2[k].is_floating_point()])
    assert np.linalg.norm(a - b) <= 1e-3 * np.linalg.norm(a)
