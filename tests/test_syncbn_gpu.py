"""att_speech.dp.enable_sync_batchnorm: two replicas (two processes on the one GPU of the test box,
gloo carrying the device tensors) that share their batch statistics reproduce the
single-process BatchNorm + Hardtanh over the whole batch — output, input gradient, parameter
gradients (summed over the replicas, as the gradient all-reduce does) and running statistics —
which is what the reference, a single process, computes (deep_speech_2.py:21,60-73)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 32, 40, 17, generator=g) * 2 + 0.5
    dy = torch.randn(6, 32, 40, 17, generator=g)
    return x, dy, [0, 1, 2, 3], [4, 5]          # uneven shards: the element counts ride along


def _run(x, dy, with_sums, time_major):
    """BatchNorm2d(32) + Hardtanh(0, 20) through the fused kernels on x (channels-last bf16, as
    the convolutions hand it over), gradient dy; returns numpy copies."""
    from att_speech.modules.encoders.native_bn import bn_hardtanh
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(32).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    act = torch.nn.Hardtanh(0, 20)
    xb = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sums = None
    if with_sums:
        xd = xb.detach().double()
        sums = torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))])
    y = bn_hardtanh(xb, bn, act, out_bf16=False, time_major=time_major, chan_sums=sums)
    g = dy.to(dev)
    if time_major:
        g = g.permute(2, 0, 1, 3).contiguous()
    y.backward(g)
    if time_major:
        y = y.permute(1, 2, 0, 3)
    return dict(y=y.detach().float().cpu().numpy(), dx=xb.grad.float().cpu().numpy(),
                dgamma=bn.weight.grad.cpu().numpy(), dbeta=bn.bias.grad.cpu().numpy(),
                rm=bn.running_mean.cpu().numpy(), rv=bn.running_var.cpu().numpy())


def _worker(rank, world, port, with_sums, time_major, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'pytorch-asr_amd'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from att_speech.dp import enable_sync_batchnorm
    enable_sync_batchnorm(True)
    x, dy, i0, i1 = _case()
    idx = i0 if rank == 0 else i1
    out = _run(x[idx], dy[idx], with_sums, time_major)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('with_sums,time_major', [(True, False), (False, False), (True, True)])
def test_two_replicas_with_shared_statistics_equal_one_process(with_sums, time_major):
    x, dy, i0, i1 = _case()
    want = _run(x, dy, with_sums, time_major)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, with_sums, time_major, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    y = np.concatenate([got[0]['y'], got[1]['y']])
    dx = np.concatenate([got[0]['dx'], got[1]['dx']])
    np.testing.assert_allclose(y, want['y'], rtol=1e-5, atol=1e-5)
    # dx is stored as bf16 (the convolution's gradient operand)
    np.testing.assert_allclose(dx, want['dx'], rtol=2e-2, atol=1e-4)
    for k in ('dgamma', 'dbeta'):
        np.testing.assert_allclose(got[0][k] + got[1][k], want[k], rtol=1e-4, atol=1e-3)
    for r in (0, 1):
        np.testing.assert_allclose(got[r]['rm'], want['rm'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got[r]['rv'], want['rv'], rtol=1e-4, atol=1e-6)
    # and it matters: without the switch a replica's output differs from the whole-batch one
    solo = _run(x[i1], dy[i1], with_sums, time_major)
    assert np.abs(solo['y'] - want['y'][4:]).max() > 1e-2
