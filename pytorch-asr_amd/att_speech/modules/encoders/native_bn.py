"""nn.BatchNorm2d + nn.Hardtanh on the MI355X as one autograd function over the
fused kernels of csrc/bnact.hip (include/asr_amd.h: asr_bn_act_{fwd,bwd}_f32).
The modules keep their parameters and buffers (state_dict keys unchanged:
conv.{1,4}.batch_norm.{weight,bias,running_mean,running_var,num_batches_tracked});
running statistics follow nn.BatchNorm2d in training mode."""
import torch
import torch.distributed as dist

from att_speech import _native

# Batch statistics shared by the replicas of a data-parallel job (att_speech.dp.
# enable_sync_batchnorm): the single-process reference normalises over the whole batch
# (deep_speech_2.py:21,60-73), per-replica statistics are a deviation DDP-style training
# accepts and this switch removes.  Two tiny all-reduces per BatchNorm layer and direction:
# the (sum, sum of squares) the convolution's epilogue left in the forward pass, the two
# gradient sums in the backward pass; the element counts ride along.
SYNC = {'on': False, 'group': None}


def _sync_active():
    return (SYNC['on'] and dist.is_available() and dist.is_initialized()
            and dist.get_world_size(SYNC['group']) > 1)


def _allreduce_sums(sums, n_local):
    """sums (f64 device tensor, any shape) += the other replicas'; returns the total count"""
    buf = torch.cat([sums.reshape(-1), sums.new_tensor([n_local])])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=SYNC['group'])
    sums.copy_(buf[:-1].view_as(sums))
    return float(buf[-1])


class BNHardtanhFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, conv_bias, running_mean, running_var, training, momentum,
                eps, lo, hi, out_bf16, time_major, chan_sums=None, sync_counts=None):
        out, mean, invstd = _native.bn_act_fwd(
            x, gamma.detach(), beta.detach(), running_mean, running_var, training, momentum,
            eps, lo, hi, out_bf16=out_bf16, time_major=time_major,
            conv_bias=None if conv_bias is None else conv_bias.detach(), chan_sums=chan_sums)
        if sync_counts is not None and running_var is not None and momentum:
            # the kernel's unbiased-variance factor n / (n - 1) used this replica's count
            nl, nt = sync_counts
            var = invstd.double().pow(-2) - eps
            running_var.add_((momentum * var * (nt / (nt - 1.0) - nl / (nl - 1.0))).float())
        ctx.has_cb = conv_bias is not None
        ctx.save_for_backward(x, gamma, beta, mean, invstd,
                              conv_bias if conv_bias is not None else gamma)
        ctx.cfg = (training, lo, hi, time_major)
        ctx.sync = training and _sync_active()
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, cb = ctx.saved_tensors
        training, lo, hi, time_major = ctx.cfg
        dx, dgamma, dbeta, dcb = _native.bn_act_bwd(
            x, gamma.detach(), beta.detach(), mean, invstd, training, lo, hi, dy,
            time_major=time_major, conv_bias=cb.detach() if ctx.has_cb else None,
            sync=_allreduce_sums if ctx.sync else None)
        return (dx, dgamma, dbeta, dcb) + (None,) * 11


def bn_hardtanh(x, bn, act, out_bf16=False, time_major=False, conv_bias=None, chan_sums=None):
    """x [B,C,H,W] f32 GPU tensor -> Hardtanh(BatchNorm2d(x + conv_bias)) as f32 / bf16,
    [B,C,H,W] or time-major [H,B,C,W]; conv_bias [C] is the bias of the convolution
    that produced x when it was run without it; chan_sums [2, C] f64: that convolution's
    per-channel (sum, sum of squares) of x, which saves the statistics pass in training."""
    use_batch_stats = bn.training or bn.running_mean is None
    momentum = 0.0
    rm = rv = None
    if use_batch_stats and bn.training and bn.track_running_stats and bn.running_mean is not None:
        bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        rm, rv = bn.running_mean, bn.running_var
    elif not use_batch_stats:
        rm, rv = bn.running_mean, bn.running_var
    if x.dtype != torch.bfloat16:
        x = x.float()
    sync_counts = None
    if use_batch_stats and bn.training and _sync_active():
        # global statistics: all-reduce the per-channel (sum, sum of squares) and scale them to
        # this replica's element count, which is what the kernel divides by
        n_local = float(x.shape[0] * x.shape[2] * x.shape[3])
        if chan_sums is None:
            xd = x.double()          # (the convolution's bias enters as a shift of the mean)
            chan_sums = torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))])
        else:
            chan_sums = chan_sums.clone()
        n_total = _allreduce_sums(chan_sums, n_local)
        chan_sums.mul_(n_local / n_total)
        sync_counts = (n_local, n_total)
    return BNHardtanhFunction.apply(x, bn.weight, bn.bias, conv_bias, rm, rv, use_batch_stats,
                                    momentum, bn.eps, float(act.min_val), float(act.max_val),
                                    out_bf16, time_major, chan_sums if use_batch_stats else None,
                                    sync_counts)
