"""Utterance-sharded data parallelism for the training step (new functionality:
the reference is single-process).

One process per GPU.  Every rank holds a replica, runs forward/backward on its
own shard of utterances (each shard re-sorted by length, descending, as the
lattice scan requires) and the gradients are SUMMED across ranks with one
all-reduce over a single flat fp32 bucket: the loss is a sum over utterances
(advanced_decoder.py:524-527), so gradients add — no averaging.  After backward
the parameter gradients are gathered into the bucket with one multi-tensor copy
(`FlatGradBucket.gather`: autograd accumulates into its own tensors) and from
then on ARE views into it: clipping, the collective and the optimizer work on the
flat buffer.  Backend: 'nccl' (= RCCL over xGMI on ROCm) for GPU tensors, 'gloo'
in the CPU tests.  The all-reduce is NOT overlapped with backward: the persistent
BiLSTM kernels need every workgroup of a team resident and nothing else on the
device while they run (README "Limits"), so the collective follows the backward
pass (0.4 ms of a 21 ms step at 27 MB over xGMI).
"""
import torch
import torch.distributed as dist


class FlatGradBucket(object):
    """All parameter gradients of `params` as views into one flat buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        # keeps the views alive (optimizer.zero_grad(set_to_none=True) would drop them)
        self.flat.zero_()

    def detach_grads(self):
        """Before backward: drop the .grad views so that autograd hands every parameter its
        gradient as a tensor of its own instead of adding it into a zeroed view — one small
        `add` kernel per parameter (27 per step for the WSJ model) — and `gather` copies them
        into the flat buffer with one multi-tensor launch."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """After backward: the gradients into the flat buffer (one `_foreach_copy_`), .grad = the
        views again; parameters that received no gradient read as zero."""
        views, grads, off = [], [], 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                views.append(view)
                grads.append(p.grad.detach())
            p.grad = view
        if views:
            torch._foreach_copy_(views, grads)

    def check_views(self):
        """backward accumulates in place into .grad when it exists; re-attach a
        view if something replaced it."""
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += p.numel()

    def all_reduce_sum(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self.check_views()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)


def shard_batch(lengths, world_size):
    """Global batch (indices sorted by length, descending) -> per-rank index
    lists, dealt in snake order (0..W-1, W-1..0, ...) so every shard has about
    the same total number of frames and a similar longest utterance; each list
    stays sorted descending (fst_utils.py:382,432, deep_speech_2.py:152 hold
    per shard)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world_size)]
    for pos, i in enumerate(order):
        rnd, k = divmod(pos, world_size)
        shards[k if rnd % 2 == 0 else world_size - 1 - k].append(i)
    return shards


def take_shard(features, feature_lens, texts, text_lens, indices):
    """The rank's slice of a collated batch (`indices` from `shard_batch`): rows picked in
    shard order, features trimmed to the shard's longest utterance — the encoder and the
    lattice scan require `features.size(1) == feature_lens[0]` (deep_speech_2.py:152,
    fst_utils.py:432) — and labels to its longest transcript."""
    idx = torch.as_tensor(indices, dtype=torch.long)
    lens, tlens = feature_lens[idx], text_lens[idx]
    return (features[idx][:, :int(lens.max())].contiguous(), lens,
            texts[idx][:, :max(1, int(tlens.max()))].contiguous(), tlens)


def broadcast_parameters(module, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def enable_sync_batchnorm(on=True, group=None):
    """Batch statistics of the fused BatchNorm + Hardtanh (csrc/bnact.hip) over ALL replicas, as
    in the single-process reference (deep_speech_2.py:21,60-73), instead of per replica: two
    small all-reduces per BatchNorm layer and direction.  Off by default, like
    DDP without SyncBatchNorm."""
    from att_speech.modules.encoders import native_bn
    native_bn.SYNC.update(on=bool(on), group=group)


def train_step(model, optimizer, batch_args, hooks=(), bucket=None, current_iteration=0,
               group=None, forward=None, fused=None):
    """One training step in the reference's order (trainer.py:229-272):
    pre_train_forward hooks -> forward -> pre_backward hooks -> zero grads ->
    backward -> [gradient all-reduce over `bucket`] -> post_backward hooks (they
    see the GLOBAL gradient, e.g. GradientClipping) -> optimizer.step() unless a
    hook asked to skip -> post_optimizer_step hooks.  `batch_args` are the
    arguments of `model.forward` (or of `forward` if given).  Returns
    (loss_dict, skipped).

    `fused` (an `att_speech.fused_step.FusedClipAdam` over `bucket`): the GradientClipping
    hook's clip / skip decision and the Adam update are taken on the device, with no read-back
    in the step; `skipped` then only reports what the other hooks asked for, the device's
    decisions arrive later through `fused.poll()` / `fused.drain()`, and a timed-out LSTM
    hand-off (whose step the device has skipped on every rank) is raised when its statistics
    arrive, a few steps late."""
    for h in hooks:
        h.pre_train_forward(model=model, optimizer=optimizer,
                            current_iteration=current_iteration)
    fwd = forward if forward is not None else model      # e.g. a wrapper that times the call
    loss_dict = fwd(*batch_args[0], **batch_args[1]) if isinstance(batch_args, tuple) \
        else fwd(**batch_args)
    loss = loss_dict['loss']
    skip = False
    for h in hooks:        # `skip or hook(...)`: hooks behind one that asked to skip are not called (trainer.py:241-248)
        skip = skip or bool(h.pre_backward(model=model, optimizer=optimizer,
                                           current_iteration=current_iteration, loss=loss))
    if bucket is not None:
        bucket.detach_grads()
    else:
        optimizer.zero_grad()
    loss.backward()
    err_word = None
    if loss.is_cuda:
        from att_speech import _native
        err_word = _native.lstm_error_word(loss.device)     # device int32[1], None if no recurrence ran
    if bucket is not None:
        bucket.gather()
        bucket.all_reduce_sum(group)
        if err_word is not None and dist.is_available() and dist.is_initialized() \
                and dist.get_world_size(group) > 1:
            # a timed-out hand-off poisons ONE rank's gradient with NaN, and the all-reduce has
            # just spread it: every rank must learn of it and discard the step together
            dist.all_reduce(err_word, op=dist.ReduceOp.MAX, group=group)
    if fused is not None:
        from att_speech import _native
        from att_speech.modules.hooks.gradient_clipping import GradientClipping
        assert bucket is not None and fused.bucket is bucket
        for h in hooks:        # every hook but the clipping one, which the device step stands for
            if not isinstance(h, GradientClipping):
                skip = skip or bool(h.post_backward(model=model, optimizer=optimizer,
                                                    current_iteration=current_iteration, loss=loss))
        if not skip:
            fused.step(err_word)
        for h in hooks:
            h.post_optimizer_step(model=model, optimizer=optimizer,
                                  current_iteration=current_iteration, loss=loss)
        for rec in fused.poll():
            if rec[3]:
                _native.lstm_raise_error(err_word)
        return loss_dict, skip
    if err_word is not None:
        # ONE read-back per step: the gradient norm the clipping hook wants and the error word
        if bucket is not None:
            pair = torch.stack([bucket.flat.norm(2), err_word[0].to(bucket.flat.dtype)]).tolist()
            bucket.cached_norm = pair[0]
            failed = pair[1] != 0
        else:
            failed = int(err_word.item()) != 0
        if failed:
            _native.lstm_raise_error(err_word)
    for h in hooks:
        if hasattr(h, 'bucket'):
            h.bucket = bucket
        skip = skip or bool(h.post_backward(model=model, optimizer=optimizer,
                                            current_iteration=current_iteration, loss=loss))
    if bucket is not None:
        bucket.cached_norm = None
    if not skip:
        optimizer.step()
    for h in hooks:
        h.post_optimizer_step(model=model, optimizer=optimizer,
                              current_iteration=current_iteration, loss=loss)
    return loss_dict, skip
