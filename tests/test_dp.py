"""Data-parallel plumbing on CPU (gloo, world_size 2): gradients of the sharded
step equal the SUM of the per-shard gradients (the loss is a sum over
utterances), shards stay length-sorted."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 4))


def _loss(model, x, lens):
    # a sum over utterances of a length-masked score, like the lattice loss
    y = model(x)                                    # [B, T, 4]
    mask = (torch.arange(x.size(1))[None, :] < lens[:, None]).float()
    return (y.logsumexp(-1) * mask).sum()


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'pytorch-asr_amd'))
    from att_speech.dp import FlatGradBucket, broadcast_parameters, shard_batch
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _toy_model()
    if rank == 1:                                   # replicas start different...
        for p in model.parameters():
            p.data.add_(1.0)
    broadcast_parameters(model)                     # ...and are made identical
    bucket = FlatGradBucket(model.parameters())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 10, 8, generator=g)
    lens = torch.tensor([10, 9, 7, 7, 4, 2])
    idx = shard_batch(lens.tolist(), world)[rank]
    for _ in range(2):                              # second pass: views survive zero_()
        bucket.zero_()
        _loss(model, x[idx], lens[idx]).backward()
        bucket.all_reduce_sum()
    grads = [p.grad.detach().numpy().copy() for p in model.parameters()]
    # global-norm clipping after the all-reduce: same decision on every rank
    from att_speech.dp import train_step
    from att_speech.modules.hooks import GradientClipping

    class Wrap(torch.nn.Module):
        def __init__(self, net):
            super(Wrap, self).__init__()
            self.net = net

        def forward(self, x, lens):
            return {'loss': _loss(self.net, x, lens)}

    hook = GradientClipping(clip_norm=0.5)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    train_step(Wrap(model), opt, ((x[idx], lens[idx]), {}), hooks=[hook], bucket=bucket)
    clip_info = (float(bucket.flat.norm()), hook.gstats.norms[0])
    # plain numpy through the queue: shared-memory tensors need the sender alive until received
    q.put((rank, idx, grads, clip_info))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradients_are_the_sum_over_shards():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # single-process reference on the whole batch
    model = _toy_model()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 10, 8, generator=g)
    lens = torch.tensor([10, 9, 7, 7, 4, 2])
    _loss(model, x, lens).backward()
    want = [p.grad for p in model.parameters()]
    total = float(torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters())))
    for rank, idx, grads, (clipped_norm, unclipped) in res:
        assert abs(unclipped - total) <= 1e-4 * total          # the hook saw the GLOBAL norm
        assert abs(clipped_norm - 0.5) <= 1e-4
        assert lens[idx].tolist() == sorted(lens[idx].tolist(), reverse=True)
        for a, b in zip(grads, want):
            torch.testing.assert_close(torch.from_numpy(a), b, rtol=1e-5, atol=1e-6)
    assert sorted(res[0][1] + res[1][1]) == list(range(6))


def test_shard_batch_balances_frames():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'pytorch-asr_amd'))
    from att_speech.dp import shard_batch
    lens = [1000 - 8 * b for b in range(64)]
    shards = shard_batch(lens, 8)
    tot = [sum(lens[i] for i in s) for s in shards]
    assert max(tot) - min(tot) <= 8 * 8
    assert all(len(s) == 8 for s in shards)


# --------------------------------------------------------------------------------------
# the real model through the real step driver: SpeechModel(DeepSpeech2 + a CTC decoder that
# has a CPU path) + FlatGradBucket + GradientClipping, two gloo ranks
# --------------------------------------------------------------------------------------
class CpuCtcDecoder(torch.nn.Module):
    """Decoder with the reference's decoder surface and a loss that runs on the CPU
    (torch's CTC; the product decoders need the MI355X): sum over utterances, like
    FSTDecoder (advanced_decoder.py:524-527)."""

    def __init__(self, sample_batch, num_classes, vocabulary=None, **kwargs):
        super(CpuCtcDecoder, self).__init__()
        self.fc = torch.nn.Linear(sample_batch['features'].size(2), num_classes)

    def forward(self, encoded, encoded_lens, texts, text_lens, spkids=None, **kwargs):
        lp = torch.log_softmax(self.fc(encoded), -1)
        loss = torch.nn.functional.ctc_loss(lp, texts.long(), encoded_lens.long(), text_lens.long(),
                                            reduction='sum')
        return {'loss': loss}


ENC_SMALL = dict(class_name='att_speech.modules.encoders.DeepSpeech2',
                 conv_kernel_sizes=[[7, 7], [7, 7]], conv_strides=[[1, 2], [3, 1]],
                 rnn_hidden_size=32, rnn_nb_layers=2, rnn_normalization='none')


def _speech_batch():
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(6, 60, 40, 1, generator=g)
    lens = torch.tensor([60, 57, 51, 45, 39, 30], dtype=torch.int32)
    for i, n in enumerate(lens.tolist()):
        feats[i, n:] = 0                 # collated batches are zero-padded (data_loader.py:32-65)
    texts = torch.randint(1, 9, (6, 4), generator=g, dtype=torch.int32)
    tlens = torch.tensor([4, 4, 3, 3, 2, 1], dtype=torch.int32)
    return feats, lens, texts, tlens


def _speech_model():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'pytorch-asr_amd'), os.path.join(root, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from att_speech.models import SpeechModel
    torch.manual_seed(3)
    feats, lens, _, _ = _speech_batch()
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    model = SpeechModel(ENC_SMALL, dict(class_name='test_dp.CpuCtcDecoder'), sb, 10,
                        [str(i) for i in range(10)])
    for mod in model.modules():          # per-replica batch statistics are a documented
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):   # deviation: keep them out
            mod.eval()
    return model


def _speech_worker(rank, world, port, q):
    model = _speech_model()
    from att_speech.dp import (FlatGradBucket, broadcast_parameters, shard_batch, take_shard,
                               train_step)
    from att_speech.modules.hooks import GradientClipping
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    broadcast_parameters(model)
    bucket = FlatGradBucket(model.parameters())
    feats, lens, texts, tlens = take_shard(*_speech_batch(), shard_batch(
        _speech_batch()[1].tolist(), world)[rank])
    hook = GradientClipping(clip_norm=1.0, skip_step_norm=1e9)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    out, skipped = train_step(model, opt, ((feats, lens, None, texts, tlens), {}),
                              hooks=[hook], bucket=bucket)
    q.put((rank, float(out['loss']), hook.gstats.norms[0], float(bucket.flat.norm()),
           bucket.flat.detach().numpy().copy(), bool(skipped)))
    dist.barrier()
    dist.destroy_process_group()


def test_speech_model_train_step_two_ranks():
    """SpeechModel + FlatGradBucket + GradientClipping through dp.train_step on two gloo ranks:
    the all-reduced gradient is the single-process gradient of the whole batch, the hook sees
    its GLOBAL norm on both ranks, and the clipped bucket has norm clip_norm."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_speech_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _speech_model()
    feats, lens, texts, tlens = _speech_batch()
    loss = model(feats, lens, None, texts, tlens)['loss']
    loss.backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad])
    total = float(flat.norm())
    assert abs(sum(r[1] for r in res) - float(loss)) <= 1e-4 * abs(float(loss))
    for rank, _, seen_norm, clipped_norm, got, skipped in res:
        assert not skipped
        assert abs(seen_norm - total) <= 1e-3 * total
        assert abs(clipped_norm - 1.0) <= 1e-3
        want = flat * (1.0 / (total + 1e-6))
        torch.testing.assert_close(torch.from_numpy(got), want, rtol=2e-3, atol=2e-5)

def test_bucket_gather_equals_in_place_accumulation():
    """FlatGradBucket.detach_grads / gather (gradients copied into the flat buffer with one
    multi-tensor launch) against backward accumulating into zeroed views; a parameter that
    receives no gradient reads as zero."""
    import torch
    from att_speech.dp import FlatGradBucket
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.ones(4))
    params = list(net.parameters()) + [unused]
    x = torch.randn(6, 5)
    bucket = FlatGradBucket(params)
    bucket.zero_()
    net(x).pow(2).sum().backward()
    want = bucket.flat.clone()
    bucket.flat.fill_(123.0)                 # stale contents must not survive
    bucket.detach_grads()
    net(x).pow(2).sum().backward()
    bucket.gather()
    assert torch.equal(bucket.flat, want)
    off = 0
    for p in params:
        assert p.grad.data_ptr() == bucket.flat[off:off + p.numel()].data_ptr()
        off += p.numel()
    assert float(unused.grad.abs().sum()) == 0.0
