"""The step boundary on the device (new functionality; the reference decides on the host:
trainer.py:250-275 calls the GradientClipping hook — modules/hooks/gradient_clipping.py:13-53,
`clip_grad_norm_` + a `float()` of the norm — and then `optimizer.step()`).

`FusedClipAdam` takes the clip / skip decision of that hook and the `torch.optim.Adam` update in
two launches over the flat gradient bucket of `att_speech.dp` (csrc/optim.hip:
asr_grad_sumsq_partials_f32, asr_adam_clip_step_f32): norm, clip factor, skip flag (norm above
`skip_step_norm`, not finite, or the persistent LSTM's error word set — MAX-reduced over the ranks
by `dp.train_step`) stay in device memory, and what the hook logs (norm, clipped, skipped) comes
back through a pinned ring that the host reads steps later, without waiting.  With no read-back
between backward and the update the host queues the next step's forward while the GPU still
works on this one: the 0.3-0.5 ms the GPU used to idle at every step boundary are gone.

Same arithmetic as the hook + torch.optim.Adam (amsgrad / maximize off, weight_decay as L2):
tests/test_fused_step_gpu.py compares parameters, moments and decisions with them over several
steps, including clipped, skipped, non-finite and LSTM-error steps.  `export_state()` writes the
moments into a torch.optim.Adam's state so that the reference's checkpointer saves what it
always saved."""
import ctypes

import numpy as np
import torch

from att_speech import _native


class FusedClipAdam(object):
    RING = 64          # steps whose statistics may be in flight

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip_norm=float('inf'), skip_step_norm=float('inf'), clipping_hook=None):
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.clip_norm, self.skip_step_norm = float(clip_norm), float(skip_step_norm)
        self.hook = clipping_hook
        flat = bucket.flat
        if not flat.is_cuda or flat.dtype != torch.float32:
            raise _native.NativeLibraryError('FusedClipAdam needs a float32 gradient bucket on the GPU')
        dev = flat.device
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.nparts = 1024
        self.partials = torch.empty(self.nparts, dtype=torch.float32, device=dev)
        self.step_words = torch.zeros(2, dtype=torch.int32, device=dev)
        self.calls = 0
        self.stats_dev = torch.zeros(self.RING, 4, dtype=torch.float32, device=dev)
        self.stats_host = torch.zeros(self.RING, 4, dtype=torch.float32).pin_memory()
        self.pending = []          # (call index, event)
        self.history = []          # (norm, clipped, skipped, err) of completed steps, in order
        self._ptrs = None
        self._chunks = None

    @classmethod
    def from_optimizer(cls, optimizer, bucket, clipping_hook=None):
        """Hyper-parameters of a torch.optim.Adam (one parameter group) and the thresholds of a
        GradientClipping hook, whose statistics keep being fed (a few steps late)."""
        if len(optimizer.param_groups) != 1:
            raise NotImplementedError('one parameter group')
        g = optimizer.param_groups[0]
        if g.get('amsgrad') or g.get('maximize'):
            raise NotImplementedError('amsgrad / maximize')
        kw = {}
        if clipping_hook is not None:
            kw = dict(clip_norm=clipping_hook.clip_norm, skip_step_norm=clipping_hook.skip_step_norm,
                      clipping_hook=clipping_hook)
        return cls(bucket, lr=g['lr'], betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'], **kw)

    def _chunk_table(self):
        """Device table of (address inside the parameter, offset in the flat buffers, count):
        rebuilt when a parameter's storage moved."""
        ptrs = tuple(p.data_ptr() for p in self.bucket.params)
        if ptrs == self._ptrs:
            return self._chunks
        ce = _native.lib().asr_adam_chunk_elems()
        rows, off = [], 0
        for p in self.bucket.params:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise NotImplementedError('contiguous float32 parameters')
            n, base = p.numel(), p.data_ptr()
            for s in range(0, n, ce):
                rows.append((base + 4 * s, off + s, min(ce, n - s)))
            off += n
        tab = np.zeros(len(rows), dtype=np.dtype([('param', '<u8'), ('off', '<u4'), ('cnt', '<u4')]))
        tab['param'] = [r[0] for r in rows]
        tab['off'] = [r[1] for r in rows]
        tab['cnt'] = [r[2] for r in rows]
        self._chunks = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(self.bucket.flat.device)
        self._nchunks = len(rows)
        self._ptrs = ptrs
        return self._chunks

    @torch.no_grad()
    def step(self, err_word=None):
        """Queue norm + update for the gradients in the bucket; nothing is read back."""
        L, p, st = _native.lib(), _native._p, _native._stream()
        flat = self.bucket.flat
        chunks = self._chunk_table()
        k = self.calls
        slot = k % self.RING
        if len(self.pending) >= self.RING - 1:      # the ring is full: wait for the oldest
            self.pending[0][1].synchronize()
            self.poll()
        _native.check(L.asr_grad_sumsq_partials_f32(p(flat), flat.numel(), p(self.partials), self.nparts, st),
                      'asr_grad_sumsq_partials_f32')
        sin, sout = self.step_words[k & 1:], self.step_words[(k + 1) & 1:]
        _native.check(L.asr_adam_clip_step_f32(
            p(chunks), self._nchunks, p(flat), p(self.m), p(self.v), p(self.partials), self.nparts,
            p(err_word) if err_word is not None else None, self.lr, self.betas[0], self.betas[1], self.eps,
            self.weight_decay, self.clip_norm, self.skip_step_norm, p(sin), p(sout),
            p(self.stats_dev[slot]), st), 'asr_adam_clip_step_f32')
        self.stats_host[slot].copy_(self.stats_dev[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((k, ev))
        self.calls += 1

    def poll(self):
        """Statistics of the steps that have completed since the last call, oldest first, as
        (norm, clipped, skipped, lstm_error); also fed to the clipping hook's log."""
        done = []
        while self.pending and self.pending[0][1].query():
            k, _ = self.pending.pop(0)
            norm, clipped, skipped, err = self.stats_host[k % self.RING].tolist()
            rec = (norm, bool(clipped), bool(skipped), bool(err))
            done.append(rec)
            self.history.append(rec)
            if self.hook is not None:
                from att_speech.modules.hooks.gradient_clipping import _NormStats
                if self.hook.gstats is None:
                    self.hook.gstats = _NormStats()
                self.hook.gstats.add(norm, int(rec[1]), int(rec[2]))
        return done

    def drain(self):
        """Wait for everything queued; returns the statistics of all steps so far."""
        if self.pending:
            self.pending[-1][1].synchronize()
        self.poll()
        return list(self.history)

    @property
    def steps_taken(self):
        """Adam's step count (a read-back: not for the training loop)."""
        return int(self.step_words[self.calls & 1].item())

    @torch.no_grad()
    def export_state(self, optimizer):
        """The moments and the step count into `optimizer` (a torch.optim.Adam over the same
        parameters), so that a checkpoint holds the usual Adam state."""
        t = self.steps_taken
        off = 0
        for p in self.bucket.params:
            n = p.numel()
            st = optimizer.state[p]
            st['step'] = torch.tensor(float(t))
            st['exp_avg'] = self.m[off:off + n].view_as(p).clone()
            st['exp_avg_sq'] = self.v[off:off + n].view_as(p).clone()
            off += n

    @torch.no_grad()
    def import_state(self, optimizer):
        """The other way round (resuming from a reference checkpoint)."""
        off, t = 0, 0
        for p in self.bucket.params:
            n = p.numel()
            st = optimizer.state.get(p, {})
            if 'exp_avg' in st:
                self.m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                self.v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                t = int(st['step'])
            off += n
        self.step_words.fill_(t)
