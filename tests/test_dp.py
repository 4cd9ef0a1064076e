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
