"""ctypes binding of libasr_amd.so (include/asr_amd.h) for torch tensors.

PyTorch is used here only as the owner of device memory and streams: every
function takes CUDA(ROCm) tensors, passes raw device pointers + sizes + the
current HIP stream through the C ABI, and returns tensors it allocated with
torch.  There is NO CPU or eager fallback: a missing library or a non-GPU tensor
raises immediately.
"""
import ctypes
import warnings
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(     # ASR_AMD_LIB: development override (kernel A/B builds)
    'ASR_AMD_LIB', os.path.join(os.path.dirname(_HERE), 'csrc', 'libasr_amd.so'))

ASR_OK, ASR_EINVAL, ASR_EUNSUPPORTED, ASR_ELAUNCH = 0, 1, 2, 3
ABI_VERSION = 22

_lib = None
# bench.py sets this to a list to collect (start, end) torch.cuda.Event pairs
# around every lattice forward-backward call (launched on the current stream)
EVENT_HOOK = None
_WARNED = {}

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

_SIGNATURES = {
    'asr_abi_version': (ctypes.c_int, []),
    'asr_strerror': (ctypes.c_char_p, [_i]),
    'asr_lattice_fwbw_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'asr_lattice_fwbw_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp,
                                  _vp, _i64, _vp]),
    'asr_lattice_fwbw_signed_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp,
                                         _vp, _i64, _vp]),
    'asr_scale_rows_f32': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'asr_lattice_fwbw_band_supported': (_i, [_i] * 7),
    'asr_lattice_fwbw_band_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp,
                                       _vp, _i64, _vp, _vp, _vp, _i, _vp]),
    'asr_lattice_viterbi_workspace_bytes': (_i64, [_i, _i, _i]),
    'asr_lattice_forward_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                     _i, _i, _i, _f, _i, _vp, _vp, _vp, _i64,
                                     _vp]),
    'asr_log_softmax_fwd_f32': (_i, [_vp, _i64, _i, _vp, _vp]),
    'asr_log_softmax_bwd_f32': (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    'asr_sub_rowmax_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'asr_argmax_rows_f32': (_i, [_vp, _i64, _i, _vp, _vp]),
    'asr_conv7x7c32_workspace_bytes': (_i64, []),
    'asr_conv7x7c32_fwd_bf16': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    'asr_conv7x7c32_bwd_data_bf16': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    'asr_conv7x7c32_wgrad_bf16': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    'asr_conv1_7x7s2_workspace_bytes': (_i64, []),
    'asr_conv1_7x7s2_fwd': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    'asr_conv1_7x7s2_wgrad': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    'asr_conv1c_7x7s2_workspace_bytes': (_i64, [_i]),
    'asr_conv1c_7x7s2_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    'asr_conv1c_7x7s2_wgrad': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    'asr_log_softmax_shift_fwd_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'asr_log_softmax_shift_bwd_f32': (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    'asr_sum_leading_f32': (_i, [_vp, _i, _i64, _vp, _vp]),
    'asr_adam_chunk_elems': (_i, []),
    'asr_grad_sumsq_partials_f32': (_i, [_vp, _i64, _vp, _i, _vp]),
    'asr_adam_clip_step_f32': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _f, _f,
                                    _vp, _vp, _vp, _vp]),
    'asr_split_bf16_f32': (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    'asr_log_softmax_shift_bwd_split_blocks': (_i, [_i64]),
    'asr_log_softmax_shift_bwd_split_bf16': (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp]),
    'asr_tcn_attention_step_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp,
                                        _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'asr_beam_step_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'asr_ctc_graph_build': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f] + [_vp] * 8),
    'asr_lattice_grouped_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'asr_lattice_grouped_fwbw_f32': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i] + [_vp] * 8 +
                                     [_f, _vp, _vp, _vp, _vp, _i64, _vp]),
    'asr_lattice_grouped_fwbw_acc_f32': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i] + [_vp] * 8 +
                                         [_f, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    'asr_lattice_grouped_forward_f32': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i] + [_vp] * 8 +
                                        [_f, _i, _vp, _vp, _vp, _i64, _vp]),
    'asr_lstm_workspace_bytes': (_i64, [_i, _i]),
    'asr_lstm_bidir_fwd_bf16': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'asr_lstm_bidir_bwd_bf16': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'asr_lstm_fused_supported': (_i, [_i, _i, _i]),
    'asr_lstm_wgrad_supported': (_i, [_i]),
    'asr_lstm_dgrad_supported': (_i, [_i]),
    'asr_lstm_dgrad_bf16': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'asr_lstm_wgrad_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'asr_lstm_wgrad_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    'asr_lstm_bidir_fwd_fused_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'asr_bn_act_workspace_bytes': (_i64, [_i]),
    'asr_bn_act_fwd_f32': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f,
                                _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    'asr_bn_act_bwd_f32': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f,
                                _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'asr_bn_act_bwd_phase_f32': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f,
                                      _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, ctypes.c_double, _vp]),
}

# include/asr_amd_experiments.h: exported only by `make EXPERIMENTS=1` builds; bound when
# present, never required (experiments_built() tells the callers).
_EXPERIMENT_SIGNATURES = {
    'asr_lstm_bidir_bwd_fused_bf16': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp,
                                           _vp, _i64, _vp, _vp]),
    'asr_lstm_bidir_fwd_fused_sum_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                               _vp, _vp]),
}
_experiments = False


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load libasr_amd.so; fail loudly if it has not been built."""
    global _lib, _experiments
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "%s is missing: build it with `make -C pytorch-asr_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path."
                % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.asr_abi_version() != ABI_VERSION:
            raise NativeLibraryError("libasr_amd.so ABI %d != expected %d" % (
                handle.asr_abi_version(), ABI_VERSION))
        _experiments = all(hasattr(handle, name) for name in _EXPERIMENT_SIGNATURES)
        if _experiments:
            for name, (res, args) in _EXPERIMENT_SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
        _lib = handle
    return _lib


def experiments_built():
    """True when libasr_amd.so was built with EXPERIMENTS=1 (include/asr_amd_experiments.h)."""
    lib()
    return _experiments


def _need_experiments(what):
    if not experiments_built():
        raise NotImplementedError(
            '%s is an experiment (include/asr_amd_experiments.h): rebuild with '
            '`make -C pytorch-asr_amd/csrc clean all EXPERIMENTS=1`' % what)


def check(code, what):
    if code == ASR_OK:
        return
    msg = "%s: %s" % (what, lib().asr_strerror(code).decode())
    if code == ASR_EINVAL:
        # the reference raises AssertionError / ValueError on malformed input
        raise AssertionError(msg)
    if code == ASR_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeLibraryError(
            "%s must be a GPU tensor: the MI355X path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Graph(object):
    """Device-resident int32/f32 copy of the reference's 4 or 8 padded graph
    matrices (fst_utils.py:222-294,491-521)."""
    __slots__ = ('src_in', 'il_in', 'w_in', 'term', 'dst_out', 'il_out',
                 'w_out', 'Bg', 'N', 'Kin', 'Kout', 'band', 'ctc_labels')

    def __init__(self, graph_matrices, device):
        # band: every state n is entered only from {n, n-1, n-2} with weights <= 0 (the CTC
        # chain lattices): lattice_fwbw sends such graphs to asr_lattice_fwbw_band_f32.  Set by
        # build_ctc_graph, or found here on the host when the matrices arrive as CPU tensors
        # (the data workers' hand-over, fst_utils.py:491-521); the kernel re-checks per utterance.
        self.band = False
        self.ctc_labels = None      # (labels [B,Lmax] i32, lens [B] i32) when built from them (order 1)
        if graph_matrices is None:          # filled in by build_ctc_graph
            return
        gm = list(graph_matrices)
        if len(gm) not in (4, 8):
            raise AssertionError("expected 4 or 8 graph matrices")
        bg = gm[0].size(0)
        if not all(m.size(0) == bg for m in gm):              # fst_utils.py:407
            raise AssertionError("graph matrices disagree on batch size")

        def to_i(m):
            return m.to(device=device, dtype=torch.int32).contiguous()

        def to_f(m):
            return m.to(device=device, dtype=torch.float32).contiguous()
        self.src_in, self.il_in, self.w_in = to_i(gm[0]), to_i(gm[1]), to_f(gm[2])
        self.term = to_f(gm[3]).reshape(bg, -1)
        self.Bg, self.N, self.Kin = self.src_in.shape
        if len(gm) == 8:
            self.dst_out, self.il_out, self.w_out = to_i(gm[4]), to_i(gm[5]), to_f(gm[6])
            self.Kout = self.dst_out.shape[2]
            self.band = _host_band_check(gm)
        else:
            self.dst_out = self.il_out = self.w_out = None
            self.Kout = 0


def _host_band_check(gm):
    """True when the in-arcs of every state n come from {n, n-1, n-2} with weights <= 0
    (CPU tensors only: no device read-back to find out)."""
    src, w = gm[0], gm[2]
    if src.is_cuda or w.is_cuda or src.dim() != 3 or src.size(1) > 256:
        return False
    valid = w > -5e19
    d = torch.arange(src.size(1)).view(1, -1, 1) - src.long()
    return bool(((d >= 0) & (d <= 2) & (w <= 0) | ~valid).all())


# Policy state of the band kernel (csrc/lattice_band.inc): it redoes, inside the launch and ~10x
# slower, the utterances whose numbers leave what a wave-wide scale factor can hold — typically
# ALL utterances of a batch while a CTC model is in its blank-collapse phase (blank ~1, labels
# ~1e-3: alpha peaks at the first states, beta at the last, the paths that matter sit 2^-300
# below both).  The kernel counts them; the count of one call is read (without a sync: pinned
# copy + event) before a later call, and when more than a tenth of a batch was redone the next
# `_BAND_COOLDOWN` calls go to the log-domain kernel, which does not care.
_BAND_STATE = {'cool': 0, 'pending': None, 'launched': 0, 'seen': (0, 0)}
_BAND_COOLDOWN = 64
_BAND_MAX_BATCH = 1 << 30      # no cap: measured faster from 384 utterances up, level with the log-domain kernel below
_BAND_COUNTER = {}             # device -> (running device counter int32[1], pinned host copy)


def _band_counter(device):
    c = _BAND_COUNTER.get(device)
    if c is None:
        c = (torch.zeros(1, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32).pin_memory())
        _BAND_COUNTER[device] = c
        _BAND_STATE.update(pending=None, launched=0, seen=(0, 0))
    return c


def _band_policy_allows():
    st = _BAND_STATE
    pend = st['pending']
    if pend is not None and pend[1].query():
        redone_total, launched_total = int(pend[0].item()), pend[2]
        redone, batch = redone_total - st['seen'][0], launched_total - st['seen'][1]
        st['seen'] = (redone_total, launched_total)
        st['pending'] = None
        if redone * 10 > batch:
            st['cool'] = _BAND_COOLDOWN
    if st['cool'] > 0:
        st['cool'] -= 1
        return False
    return True


def _band_policy_record(device, B):
    """the kernel's running redo counter, read without a sync (pinned copy + event) once in a while"""
    st = _BAND_STATE
    st['launched'] += B
    if st['pending'] is not None:
        return
    dev_cnt, host = _band_counter(device)
    host.copy_(dev_cnt, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st['pending'] = (host, ev, st['launched'])


def scale_rows_(x, scale):
    """asr_scale_rows_f32: x [T,B,C] *= scale [B] in place; utterances whose factor is exactly 1
    are not touched."""
    x = _dev(x, torch.float32, 'x')
    scale = _dev(scale.to(torch.float32), torch.float32, 'scale')
    T, B, C = x.shape
    check(lib().asr_scale_rows_f32(_p(x), T, B, C, _p(scale), _stream()), 'asr_scale_rows_f32')
    return x


def lattice_fwbw(lp, lens, graph, neg_inf=-1e20, want_bwd_total=False, grad_sign=1.0):
    """asr_lattice_fwbw_f32: returns (logZ [B], grad [T,B,C], logZ_bwd|None); grad_sign = -1:
    the occupancies come back negated (asr_lattice_fwbw_signed_f32: the gradient of -logZ)."""
    lp = _dev(lp, torch.float32, 'log_probs')
    lens = _dev(lens, torch.int32, 'act_lens')
    T, B, C = lp.shape
    if graph.Bg not in (1, B):
        raise AssertionError("graph batch %d not in (1, %d)" % (graph.Bg, B))
    L = lib()
    logZ = torch.empty(B, dtype=torch.float32, device=lp.device)
    grad = torch.empty_like(lp)
    zb = torch.empty(B, dtype=torch.float32, device=lp.device) if want_bwd_total else None
    if (T + 64) * B * C * 4 >= 2 ** 32 and graph.N <= 512 and not _WARNED.get('fits32'):
        _WARNED['fits32'] = True        # csrc/lattice.hip: 32-bit buffer offsets in the fast kernels
        warnings.warn('lattice_fwbw: T*B*C*4 = %.2f GiB >= 4 GiB (e.g. bi-char CTC at B >= ~1100): the '
                      'meet-in-the-middle kernels address with 32-bit offsets, this call runs the '
                      'generic (much slower) kernel; split the batch' % (T * B * C * 4 / 2.0 ** 30))
    nbytes = L.asr_lattice_fwbw_workspace_bytes(T, B, C, graph.N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=lp.device)
    hook = EVENT_HOOK
    if hook is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    # band lattices (CTC chains of mono-character transcripts): the linear-domain kernel
    # (csrc/lattice_band.inc).  ASR_LATTICE_BAND: 0 never (the log-domain
    # state-labelled kernel, for A/B runs), 2 always, 1 (default) unless most of a recent batch
    # had to be redone in the log domain
    band_env = os.environ.get('ASR_LATTICE_BAND', '1')
    use_band = (graph.band and band_env != '0' and
                L.asr_lattice_fwbw_band_supported(T, B, C, graph.N, graph.Kin, graph.Kout, graph.Bg))
    if use_band and band_env != '2' and (B > _BAND_MAX_BATCH or not _band_policy_allows()):
        use_band = False
    args = (_p(lp), T, B, C, _p(lens), _p(graph.src_in), _p(graph.il_in),
            _p(graph.w_in), _p(graph.term), _p(graph.dst_out), _p(graph.il_out),
            _p(graph.w_out), graph.N, graph.Kin, graph.Kout, graph.Bg,
            float(neg_inf), _p(logZ), _p(grad), _p(zb), _p(ws), nbytes)
    sargs = args[:17] + (float(grad_sign),) + args[17:]
    if use_band:
        cl = getattr(graph, 'ctc_labels', None)
        if cl is not None and os.environ.get('ASR_BAND_LABELS', '1') == '0':
            cl = None
        check(L.asr_lattice_fwbw_band_f32(*sargs, _p(_band_counter(lp.device)[0]),
                                          _p(cl[0]) if cl else None, _p(cl[1]) if cl else None,
                                          int(cl[0].shape[1]) if cl else 0, _stream()),
              'asr_lattice_fwbw_band_f32')
    elif grad_sign != 1.0:
        check(L.asr_lattice_fwbw_signed_f32(*sargs, _stream()), 'asr_lattice_fwbw_signed_f32')
    else:
        check(L.asr_lattice_fwbw_f32(*args, _stream()), 'asr_lattice_fwbw_f32')
    if hook is not None:
        ev1.record()
        hook.append((ev0, ev1))
    if use_band and band_env != '2':
        _band_policy_record(lp.device, B)
    if use_band and os.environ.get('ASR_LATTICE_BAND_DEBUG'):
        # development aid: why utterances were redone by the in-kernel log-domain body (the last
        # word of each utterance's workspace region, csrc/lattice_band.inc); synchronises
        wc = (graph.N + 63) // 64 * 64 + 64
        why = ws[:B * (T + 2) * wc * 4].view(torch.int32).view(B, (T + 2) * wc)[:, -1].cpu()
        vals, cnts = torch.unique(why, return_counts=True)
        print('[band] B=%d T=%d N=%d fallback reasons {code: utterances} %s' % (
            B, T, graph.N, dict(zip(vals.tolist(), cnts.tolist()))), flush=True)
    return logZ, grad, zb


def lattice_forward(lp, lens, graph, neg_inf=-1e20, viterbi=False, want_path=False):
    """asr_lattice_forward_f32: returns (score [B], best_il [T,B] | None)."""
    lp = _dev(lp, torch.float32, 'log_probs')
    lens = _dev(lens, torch.int32, 'act_lens')
    T, B, C = lp.shape
    if graph.Bg not in (1, B):
        raise AssertionError("graph batch %d not in (1, %d)" % (graph.Bg, B))
    L = lib()
    score = torch.empty(B, dtype=torch.float32, device=lp.device)
    best, ws, nbytes = None, None, 0
    if viterbi and want_path:
        best = torch.empty((T, B), dtype=torch.int32, device=lp.device)
        nbytes = L.asr_lattice_viterbi_workspace_bytes(T, B, graph.N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=lp.device)
    check(L.asr_lattice_forward_f32(
        _p(lp), T, B, C, _p(lens), _p(graph.src_in), _p(graph.il_in),
        _p(graph.w_in), _p(graph.term), graph.N, graph.Kin, graph.Bg,
        float(neg_inf), int(bool(viterbi)), _p(score), _p(best), _p(ws), nbytes,
        _stream()), 'asr_lattice_forward_f32')
    return score, best


def log_softmax_fwd(x, group):
    x = _dev(x, torch.float32, 'acts')
    if x.numel() % group:
        raise AssertionError("group %d does not divide %d" % (group, x.numel()))
    y = torch.empty_like(x)
    check(lib().asr_log_softmax_fwd_f32(_p(x), x.numel() // group, group, _p(y),
                                        _stream()), 'asr_log_softmax_fwd_f32')
    return y


def log_softmax_bwd(y, dy, group):
    y = _dev(y, torch.float32, 'y')
    dy = _dev(dy, torch.float32, 'dy')
    dx = torch.empty_like(y)
    check(lib().asr_log_softmax_bwd_f32(_p(y), _p(dy), y.numel() // group, group,
                                        _p(dx), _stream()),
          'asr_log_softmax_bwd_f32')
    return dx


def sub_rowmax(x, lens):
    """asr_sub_rowmax_f32: returns (x - rowmax, row_max [T,B], max_sum [B])."""
    x = _dev(x, torch.float32, 'logits')
    lens = _dev(lens, torch.int32, 'lens')
    T, B, C = x.shape
    y = torch.empty_like(x)
    row_max = torch.empty((T, B), dtype=torch.float32, device=x.device)
    max_sum = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib().asr_sub_rowmax_f32(_p(x), T, B, C, _p(lens), _p(y), _p(row_max),
                                   _p(max_sum), _stream()), 'asr_sub_rowmax_f32')
    return y, row_max, max_sum


def argmax_rows(x):
    """asr_argmax_rows_f32 over the last axis: returns int32 indices x.shape[:-1]."""
    x = _dev(x, torch.float32, 'logits')
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
    check(lib().asr_argmax_rows_f32(_p(x), x.numel() // C, C, _p(out), _stream()),
          'asr_argmax_rows_f32')
    return out


_LSTM_ERR = {}      # device index -> int32[1] timeout word of the persistent recurrence


def _lstm_err_flag(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    f = _LSTM_ERR.get(idx)
    if f is None:
        f = _LSTM_ERR[idx] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


def lstm_error_word(device):
    """The device word a timed-out hand-off of the persistent recurrence sets (int32[1]), or None
    when no recurrence has run on `device`; `dp.train_step` folds it into its one read-back."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _LSTM_ERR.get(idx)


def lstm_raise_error(word):
    word.zero_()
    raise RuntimeError(
        'persistent BiLSTM recurrence: a team hand-off timed out (another stream\'s kernel kept a '
        'team mate off the device?); the outputs of that call, and the gradients of this step on '
        'every rank, are NaN: the step is discarded. Set ASR_LSTM_PERSIST=0 to use one launch per '
        'time step.')


def lstm_check_errors():
    """Raise if a hand-off of the persistent BiLSTM recurrence timed out since the last
    check (csrc/lstm.hip team_wait: the kernel then poisons its outputs with NaN; the
    step must be discarded).  One 4-byte read-back per device; `dp.train_step` calls it
    once per step."""
    for idx, f in _LSTM_ERR.items():
        if int(f.item()) != 0:
            f.zero_()
            raise RuntimeError(
                'persistent BiLSTM recurrence: a team hand-off timed out on cuda:%d (another '
                "stream's kernel kept a team mate off the device?); outputs of that call are "
                'NaN. Set ASR_LSTM_PERSIST=0 to use one launch per time step.' % idx)


def lstm_bidir_fwd(gx, whh_bf16, lens, want_y=True):
    """asr_lstm_bidir_fwd_bf16: gx [T,B,2,4H] f32 or bf16, whh [2,4H,H] bf16, lens [B] i32
    -> (y [T,B,2,H] f32, y_bf16 [2,T+2,B,H], gates [T,2,B,H,4] bf16, csave [T,2,B,H])."""
    gx = _dev(gx, gx.dtype if gx.dtype == torch.bfloat16 else torch.float32, 'gx')
    whh_bf16 = _dev(whh_bf16, torch.bfloat16, 'whh')
    lens = _dev(lens, torch.int32, 'lens')
    T, B, _, H4 = gx.shape
    H = H4 // 4
    L = lib()
    y = torch.empty((T, B, 2, H), dtype=torch.float32, device=gx.device) if want_y else None
    ybf = torch.empty((2, T + 2, B, H), dtype=torch.bfloat16, device=gx.device)
    gates = torch.empty((T, 2, B, H, 4), dtype=torch.bfloat16, device=gx.device)
    csave = torch.empty((T, 2, B, H), dtype=torch.float32, device=gx.device)
    nbytes = L.asr_lstm_workspace_bytes(B, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gx.device)
    check(L.asr_lstm_bidir_fwd_bf16(_p(gx), int(gx.dtype == torch.bfloat16), _p(whh_bf16),
                                    _p(lens), T, B, H, _p(y),
                                    _p(ybf), _p(gates), _p(csave), _p(ws), nbytes,
                                    _p(_lstm_err_flag(gx.device)), _stream()), 'asr_lstm_bidir_fwd_bf16')
    return y, ybf, gates, csave


def lstm_fused_supported(B, H, backward=False, F=None, dirsum=False):
    """asr_lstm_fused_supported: can asr_lstm_bidir_fwd_fused_bf16 (bit 0) /
    asr_lstm_bidir_bwd_fused_bf16 (bit 1) / asr_lstm_bidir_fwd_fused_sum_bf16 (bit 2, `dirsum`)
    run this (batch, hidden, input size)?"""
    F = H if F is None else F
    if (dirsum or backward) and not experiments_built():
        return False
    return bool(lib().asr_lstm_fused_supported(int(B), int(H), int(F)) & (4 if dirsum else 2 if backward else 1))


def lstm_bidir_fwd_fused(x_bf16, wih_bf16, whh_bf16, lens, want_y=True, want_sum=False):
    """asr_lstm_bidir_fwd_fused_bf16: x [T,B,F] bf16, wih [2*4H,F] bf16, whh [2,4H,H] bf16
    -> the outputs of lstm_bidir_fwd, the input projection computed inside the recurrence.
    want_sum: also xsum [T,B,H] bf16 = ybf[0, 1:T+1] + ybf[1, 1:T+1], written by the recurrence
    itself (asr_lstm_bidir_fwd_fused_sum_bf16); returned as a fifth value."""
    x_bf16 = _dev(x_bf16, torch.bfloat16, 'x')
    wih_bf16 = _dev(wih_bf16, torch.bfloat16, 'wih')
    whh_bf16 = _dev(whh_bf16, torch.bfloat16, 'whh')
    lens = _dev(lens, torch.int32, 'lens')
    T, B, F = x_bf16.shape
    H = whh_bf16.shape[-1]
    if tuple(wih_bf16.shape) != (8 * H, F) or tuple(whh_bf16.shape) != (2, 4 * H, H):
        raise ValueError('lstm_bidir_fwd_fused: wih must be [8H, F], whh [2, 4H, H]')
    L = lib()
    dev = x_bf16.device
    y = torch.empty((T, B, 2, H), dtype=torch.float32, device=dev) if want_y else None
    ybf = torch.empty((2, T + 2, B, H), dtype=torch.bfloat16, device=dev)
    gates = torch.empty((T, 2, B, H, 4), dtype=torch.bfloat16, device=dev)
    csave = torch.empty((T, 2, B, H), dtype=torch.float32, device=dev)
    nbytes = L.asr_lstm_workspace_bytes(B, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if want_sum:
        _need_experiments('asr_lstm_bidir_fwd_fused_sum_bf16')
        xsum = torch.empty((T, B, H), dtype=torch.bfloat16, device=dev)
        check(L.asr_lstm_bidir_fwd_fused_sum_bf16(_p(x_bf16), _p(wih_bf16), _p(whh_bf16), _p(lens), T, B, H, F,
                                                  _p(y), _p(ybf), _p(gates), _p(csave), _p(xsum), _p(ws), nbytes,
                                                  _p(_lstm_err_flag(dev)), _stream()),
              'asr_lstm_bidir_fwd_fused_sum_bf16')
        if T % 2:           # the middle frame of an odd T: both directions reach it in the first launch
            m = T // 2
            torch.add(ybf[0, m + 1], ybf[1, m + 1], out=xsum[m])
        return y, ybf, gates, csave, xsum
    check(L.asr_lstm_bidir_fwd_fused_bf16(_p(x_bf16), _p(wih_bf16), _p(whh_bf16), _p(lens), T, B, H, F,
                                          _p(y), _p(ybf), _p(gates), _p(csave), _p(ws), nbytes,
                                          _p(_lstm_err_flag(dev)), _stream()),
          'asr_lstm_bidir_fwd_fused_bf16')
    return y, ybf, gates, csave


def lens_on(lens, device):
    """int32 device copy of a CPU length vector.  A host-to-device copy in the middle of a
    step waits for everything queued before it (pageable memory), after which the small
    kernels that follow run at the host's launch rate; whoever creates the lengths early in
    the step (DeepSpeech2.forward) can attach the device copy with `attach_device_lens` and
    every later consumer picks it up here."""
    if not isinstance(lens, torch.Tensor):
        lens = torch.as_tensor(lens)
    if lens.is_cuda:
        return lens.to(device=device, dtype=torch.int32)
    hit = getattr(lens, '_asr_dev', None)
    if hit is not None and hit[0] == str(device) and hit[1] == lens._version:
        return hit[2]
    return lens.to(device=device, dtype=torch.int32)


def attach_device_lens(lens, device):
    """copy a CPU length vector to the device now and remember the copy on the tensor object"""
    if isinstance(lens, torch.Tensor) and not lens.is_cuda:
        lens._asr_dev = (str(device), lens._version, lens.to(device=device, dtype=torch.int32))
    return lens


def lstm_dgrad_supported(H):
    return bool(lib().asr_lstm_dgrad_supported(int(H)))


def lstm_dgrad(dgates, w_ih_bf16):
    """asr_lstm_dgrad_bf16: dgates [T,B,2,4H] bf16, w_ih [8H,H] bf16 -> dx [T,B,H] f32."""
    dgates = _dev(dgates, torch.bfloat16, 'dgates')
    w_ih_bf16 = _dev(w_ih_bf16, torch.bfloat16, 'w_ih')
    T, B, H = dgates.shape[0], dgates.shape[1], dgates.shape[3] // 4
    if tuple(w_ih_bf16.shape) != (8 * H, H):
        raise ValueError('lstm_dgrad: w_ih must be [8H, H]')
    dx = torch.empty((T, B, H), dtype=torch.float32, device=dgates.device)
    check(lib().asr_lstm_dgrad_bf16(_p(dgates), _p(w_ih_bf16), T, B, H, _p(dx), _stream()),
          'asr_lstm_dgrad_bf16')
    return dx


def lstm_wgrad_supported(H):
    return bool(lib().asr_lstm_wgrad_supported(int(H)))


def lstm_wgrad(dgates, x_bf16, y_bf16):
    """asr_lstm_wgrad_bf16: dgates [T,B,2,4H] bf16, x [T*B,H] bf16 or None, y_bf16 [2,T+2,B,H]
    -> (dw_ih [8H,H] f32 or None, dw_hh [2,4H,H] f32)."""
    dgates = _dev(dgates, torch.bfloat16, 'dgates')
    y_bf16 = _dev(y_bf16, torch.bfloat16, 'y_bf16')
    T, B, H = dgates.shape[0], dgates.shape[1], dgates.shape[3] // 4
    dev = dgates.device
    if x_bf16 is not None:
        x_bf16 = _dev(x_bf16, torch.bfloat16, 'x')
        if x_bf16.numel() != T * B * H:
            raise ValueError('lstm_wgrad: x must be [T*B, H]')
    L = lib()
    dw_ih = torch.empty((8 * H, H), dtype=torch.float32, device=dev) if x_bf16 is not None else None
    dw_hh = torch.empty((2, 4 * H, H), dtype=torch.float32, device=dev)
    nbytes = L.asr_lstm_wgrad_workspace_bytes(T, B, H, int(x_bf16 is not None))
    if nbytes < 0:
        raise NativeLibraryError('asr_lstm_wgrad: unsupported hidden size %d' % H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.asr_lstm_wgrad_bf16(_p(dgates), _p(x_bf16), _p(y_bf16), T, B, H, _p(dw_ih), _p(dw_hh),
                                _p(ws), nbytes, _stream()), 'asr_lstm_wgrad_bf16')
    return dw_ih, dw_hh


def _dy_mode(dy, planes):
    if planes:
        if dy.dim() != 4 or dy.shape[0] != 2:
            raise ValueError('dy planes must be [2,T,B,H]')
        return 2, dy.shape[1], dy.shape[2]
    return int(dy.dim() == 3), dy.shape[0], dy.shape[1]


def lstm_bidir_bwd(dy, whhT_bf16, lens, gates, csave, planes=False):
    """asr_lstm_bidir_bwd_bf16 -> dgates [T,B,2,4H] bf16.  dy [T,B,2,H], or [T,B,H] when
    the two directions share one gradient (their outputs are summed), or with `planes`
    [2,T,B,H]: that shared gradient as the sum of two planes (lstm_bidir_bwd_fused's dx)."""
    dy = _dev(dy, torch.float32, 'dy')
    whhT_bf16 = _dev(whhT_bf16, torch.bfloat16, 'whhT')
    lens = _dev(lens, torch.int32, 'lens')
    mode, T, B = _dy_mode(dy, planes)
    H = dy.shape[-1]
    L = lib()
    dgates = torch.empty((T, B, 2, 4 * H), dtype=torch.bfloat16, device=dy.device)
    nbytes = L.asr_lstm_workspace_bytes(B, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    check(L.asr_lstm_bidir_bwd_bf16(_p(dy), mode, _p(whhT_bf16), _p(lens), T, B, H,
                                    _p(gates), _p(csave), _p(dgates), _p(ws), nbytes,
                                    _p(_lstm_err_flag(dy.device)), _stream()),
          'asr_lstm_bidir_bwd_bf16')
    return dgates


def lstm_bidir_bwd_fused(dy, whhT_bf16, wihT_bf16, lens, gates, csave, planes=False):
    """asr_lstm_bidir_bwd_fused_bf16 -> (dgates [T,B,2,4H] bf16, dx [2,T,B,H] f32 planes whose
    sum is the layer's input gradient).  whhT, wihT: [2,H,4H] bf16; dy as in lstm_bidir_bwd."""
    dy = _dev(dy, torch.float32, 'dy')
    whhT_bf16 = _dev(whhT_bf16, torch.bfloat16, 'whhT')
    wihT_bf16 = _dev(wihT_bf16, torch.bfloat16, 'wihT')
    lens = _dev(lens, torch.int32, 'lens')
    _need_experiments('asr_lstm_bidir_bwd_fused_bf16')
    mode, T, B = _dy_mode(dy, planes)
    H = dy.shape[-1]
    if tuple(wihT_bf16.shape) != (2, H, 4 * H) or tuple(whhT_bf16.shape) != (2, H, 4 * H):
        raise ValueError('lstm_bidir_bwd_fused: the layer input size must equal the hidden size')
    L = lib()
    dgates = torch.empty((T, B, 2, 4 * H), dtype=torch.bfloat16, device=dy.device)
    dx = torch.empty((2, T, B, H), dtype=torch.float32, device=dy.device)
    nbytes = L.asr_lstm_workspace_bytes(B, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    check(L.asr_lstm_bidir_bwd_fused_bf16(_p(dy), mode, _p(whhT_bf16), _p(wihT_bf16), _p(lens), T, B, H,
                                          _p(gates), _p(csave), _p(dgates), _p(dx), _p(ws), nbytes,
                                          _p(_lstm_err_flag(dy.device)), _stream()),
          'asr_lstm_bidir_bwd_fused_bf16')
    return dgates, dx


class GroupedGraph(object):
    """Device copy of a group-factored decoding graph (include/asr_amd.h,
    asr_lattice_grouped_*): numpy/torch arrays g_of, h_of, label, selfx, uniq [N],
    mem_g [G,Wg], mem_h [G,Wh], term [N]."""

    def __init__(self, st, device):
        def i32(a):
            return torch.as_tensor(a).to(device=device, dtype=torch.int32).contiguous()
        self.g_of, self.h_of, self.label = i32(st['g_of']), i32(st['h_of']), i32(st['label'])
        self.selfx, self.uniq = i32(st['selfx']), i32(st['uniq'])
        self.mem_g, self.mem_h = i32(st['mem_g']), i32(st['mem_h'])
        self.term = torch.as_tensor(st['term']).to(device=device, dtype=torch.float32).contiguous()
        self.N = int(self.g_of.numel())
        self.G, self.Wg = self.mem_g.shape
        self.Wh = self.mem_h.shape[1]

    def _args(self):
        return (self.N, self.G, self.Wg, self.Wh, _p(self.g_of), _p(self.h_of), _p(self.label),
                _p(self.selfx), _p(self.uniq), _p(self.mem_g), _p(self.mem_h), _p(self.term))


def grouped_fwbw(lp, lens, gg, neg_inf=-1e20, want_bwd_total=False, add_to=None):
    """asr_lattice_grouped_fwbw_f32 -> (logZ [B], grad [T,B,C], logZ_bwd | None); add_to: a
    [T,B,C] f32 tensor the occupancies are ADDED to (asr_lattice_grouped_fwbw_acc_f32) and which
    is returned as grad."""
    lp = _dev(lp, torch.float32, 'log_probs')
    lens = _dev(lens, torch.int32, 'act_lens')
    T, B, C = lp.shape
    L = lib()
    logZ = torch.empty(B, dtype=torch.float32, device=lp.device)
    if add_to is not None:
        assert add_to.shape == lp.shape and add_to.is_contiguous() and add_to.dtype == torch.float32
    grad = torch.empty_like(lp) if add_to is None else add_to
    zb = torch.empty(B, dtype=torch.float32, device=lp.device) if want_bwd_total else None
    nbytes = L.asr_lattice_grouped_workspace_bytes(T, B, gg.N, gg.G)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=lp.device)
    check(L.asr_lattice_grouped_fwbw_acc_f32(
        _p(lp), T, B, C, _p(lens), *gg._args(), float(neg_inf), int(add_to is not None), _p(logZ),
        _p(grad), _p(zb), _p(ws), nbytes, _stream()), 'asr_lattice_grouped_fwbw_acc_f32')
    return logZ, grad, zb


def grouped_forward(lp, lens, gg, neg_inf=-1e20, viterbi=False, want_path=False):
    """asr_lattice_grouped_forward_f32 -> (score [B], best_il [T,B] | None)."""
    lp = _dev(lp, torch.float32, 'log_probs')
    lens = _dev(lens, torch.int32, 'act_lens')
    T, B, C = lp.shape
    L = lib()
    score = torch.empty(B, dtype=torch.float32, device=lp.device)
    best, ws, nbytes = None, None, 0
    if viterbi and want_path:
        best = torch.empty((T, B), dtype=torch.int32, device=lp.device)
        nbytes = L.asr_lattice_grouped_workspace_bytes(T, B, gg.N, gg.G)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=lp.device)
    check(L.asr_lattice_grouped_forward_f32(
        _p(lp), T, B, C, _p(lens), *gg._args(), float(neg_inf), int(bool(viterbi)), _p(score),
        _p(best), _p(ws), nbytes, _stream()), 'asr_lattice_grouped_forward_f32')
    return score, best


def build_ctc_graph(labels, label_lens, num_symbols, context_order,
                    allow_nonblank_selfloops=True, use_contextual_blanks=False,
                    nc_weight=-1e20):
    """asr_ctc_graph_build: labels [B,Lmax] i32 (GPU), label_lens [B] i32 (GPU)
    -> Graph with [B, 2*Lmax+1, 3] arc arrays built on the device."""
    labels = _dev(labels, torch.int32, 'labels')
    label_lens = _dev(label_lens, torch.int32, 'label_lens')
    B, Lmax = labels.shape
    N = 2 * Lmax + 1
    dev = labels.device
    g = Graph(None, dev)

    def ti():
        return torch.empty((B, N, 3), dtype=torch.int32, device=dev)

    def tf():
        return torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    g.src_in, g.il_in, g.w_in = ti(), ti(), tf()
    g.dst_out, g.il_out, g.w_out = ti(), ti(), tf()
    g.term = torch.empty((B, N), dtype=torch.float32, device=dev)
    g.Bg, g.N, g.Kin, g.Kout = B, N, 3, 3
    g.band = True        # the 2 L + 1 chain in natural state order, both context orders
    # mono-character chains: the band kernel writes the chain down from the transcript itself
    g.ctc_labels = (labels, label_lens) if int(context_order) == 1 and Lmax > 0 else None
    check(lib().asr_ctc_graph_build(
        _p(labels), _p(label_lens), B, Lmax, int(num_symbols), int(context_order),
        int(bool(allow_nonblank_selfloops)), int(bool(use_contextual_blanks)),
        float(nc_weight), _p(g.src_in), _p(g.il_in), _p(g.w_in), _p(g.term),
        _p(g.dst_out), _p(g.il_out), _p(g.w_out), _stream()), 'asr_ctc_graph_build')
    return g


def _nchw_or_nhwc(t, name, dtype):
    """4-D GPU tensor as stored: (tensor, channels_last flag) without a layout copy when it
    is dense in either format."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeLibraryError(
            "%s must be a GPU tensor: the MI355X path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.is_contiguous():
        return t, False
    if t.is_contiguous(memory_format=torch.channels_last):
        return t, True
    return t.contiguous(), False


def bn_act_fwd(x, gamma, beta, running_mean, running_var, training, momentum, eps, lo, hi,
               out_bf16=False, time_major=False, conv_bias=None, chan_sums=None):
    """asr_bn_act_fwd_f32 on x [B,C,H,W] f32 (NCHW or channels_last storage) or bf16
    (channels_last) ->
    (out, save_mean [C], save_invstd [C]); out is [B,C,H,W] in x's memory format or,
    time_major, a dense [H,B,C,W]; running stats are updated in place when training."""
    x, cl = _nchw_or_nhwc(x, 'x', x.dtype if x.dtype == torch.bfloat16 else torch.float32)
    gamma = _dev(gamma, torch.float32, 'gamma')
    beta = _dev(beta, torch.float32, 'beta')
    B, C, H, W = x.shape
    if cl and (C % 4 or C > 1024 or 256 % (C // 4)):
        x, cl = x.contiguous(), False
    if x.dtype == torch.bfloat16 and not cl:        # bf16 input is a channels-last path
        x = x.float()
    L = lib()
    odt = torch.bfloat16 if out_bf16 else torch.float32
    if time_major:
        out = torch.empty((H, B, C, W), dtype=odt, device=x.device)
    else:
        out = torch.empty((B, C, H, W), dtype=odt, device=x.device,
                          memory_format=torch.channels_last if cl else torch.contiguous_format)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = L.asr_bn_act_workspace_bytes(C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    if conv_bias is not None:
        conv_bias = _dev(conv_bias, torch.float32, 'conv_bias')
    check(L.asr_bn_act_fwd_f32(_p(x), int(x.dtype == torch.bfloat16), _p(conv_bias), B, C, H, W, _p(gamma), _p(beta), _p(running_mean),
                               _p(running_var), int(cl), int(bool(training)), float(momentum),
                               float(eps), float(lo), float(hi), _p(out), int(out_bf16),
                               int(time_major), _p(mean), _p(invstd), _p(chan_sums), _p(ws), nbytes,
                               _stream()),
          'asr_bn_act_fwd_f32')
    return out, mean, invstd


def bn_act_bwd(x, gamma, beta, mean, invstd, training, lo, hi, dy, time_major=False,
               conv_bias=None, sync=None):
    """asr_bn_act_bwd_f32 -> (dx [B,C,H,W] f32 in x's memory format, dgamma [C], dbeta [C],
    dconv_bias [C] | None);
    dy f32 or bf16 in the layout the forward wrote.
    sync: a callable (sums [2 C] f64 device tensor, n_local) -> n_total that adds the per-channel
    sums up over the replicas in place (SyncBN-style data parallelism): the call then runs as
    asr_bn_act_bwd_phase_f32 phase 1 (local sums + parameter gradients), sync, phase 2 (dx)."""
    x, cl = _nchw_or_nhwc(x, 'x', x.dtype if x.dtype == torch.bfloat16 else torch.float32)
    if cl and (x.shape[1] % 4 or x.shape[1] > 1024 or 256 % (x.shape[1] // 4)):
        x, cl = x.contiguous(), False
    if x.dtype == torch.bfloat16 and not cl:
        x = x.float()
    if dy.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("dy must be float32 or bfloat16")
    if time_major or not cl:
        dy = _dev(dy, dy.dtype, 'dy')
    else:
        dy = dy.contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    L = lib()
    dx = torch.empty_like(x)                      # preserves the memory format
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    dcb = None
    if conv_bias is not None:
        conv_bias = _dev(conv_bias, torch.float32, 'conv_bias')
        dcb = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = L.asr_bn_act_workspace_bytes(C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    if sync is not None and training:
        def phase(k, n_total):
            check(L.asr_bn_act_bwd_phase_f32(
                _p(x), int(x.dtype == torch.bfloat16), _p(conv_bias), B, C, H, W, _p(gamma), _p(beta),
                _p(mean), _p(invstd), int(cl), 1, float(lo), float(hi), _p(dy),
                int(dy.dtype == torch.bfloat16), int(time_major), _p(dx), _p(dgamma), _p(dbeta), _p(dcb),
                _p(ws), nbytes, k, float(n_total), _stream()), 'asr_bn_act_bwd_phase_f32')
        phase(1, 0.0)
        sums = ws[:2 * C * 8].view(torch.float64)
        phase(2, sync(sums, float(B) * H * W))
        return dx, dgamma, dbeta, dcb
    check(L.asr_bn_act_bwd_f32(_p(x), int(x.dtype == torch.bfloat16), _p(conv_bias), B, C, H, W, _p(gamma), _p(beta), _p(mean), _p(invstd),
                               int(cl), int(bool(training)), float(lo), float(hi), _p(dy),
                               int(dy.dtype == torch.bfloat16), int(time_major), _p(dx),
                               _p(dgamma), _p(dbeta), _p(dcb), _p(ws), nbytes, _stream()),
          'asr_bn_act_bwd_f32')
    return dx, dgamma, dbeta, dcb


def tcn_attention_step(eproj, enc, enc_lens, filt, glob, w_score, b_score, temperature,
                       att_prev, parent, beam):
    """asr_tcn_attention_step_f32 -> (att_new [B*beam, T], context [B*beam, E])."""
    eproj, enc = _dev(eproj, torch.float32, 'eproj'), _dev(enc, torch.float32, 'enc')
    filt, glob = _dev(filt, torch.float32, 'filt'), _dev(glob, torch.float32, 'glob')
    att_prev = _dev(att_prev, torch.float32, 'att_prev')
    w_score = _dev(w_score, torch.float32, 'w_score')
    enc_lens = _dev(enc_lens, torch.int32, 'enc_lens')
    T, B, A = eproj.shape
    E = enc.shape[2]
    hyps = B * beam
    taps = filt.numel() // (hyps * A)
    att_new = torch.empty((hyps, T), dtype=torch.float32, device=enc.device)
    ctx = torch.empty((hyps, E), dtype=torch.float32, device=enc.device)
    par = None if parent is None else _dev(parent, torch.int32, 'parent')
    check(lib().asr_tcn_attention_step_f32(
        _p(eproj), _p(enc), _p(enc_lens), _p(filt), _p(glob), _p(w_score), float(b_score),
        float(temperature), _p(att_prev), _p(par), T, B, beam, A, taps, E, _p(att_new), _p(ctx),
        _stream()), 'asr_tcn_attention_step_f32')
    return att_new, ctx


def beam_step(logits, scores_in, scores_out, est_in, est_out, step, B, beam, len_div, state):
    """asr_beam_step_f32; `state` = dict of the per-utterance device arrays (finished_count,
    best_score, best_len, best_tokens, new_input, parent, done)."""
    C = logits.shape[-1]
    check(lib().asr_beam_step_f32(
        _p(logits), _p(scores_in), _p(scores_out), _p(est_in), _p(est_out), int(step), B, beam, C,
        est_in.shape[1], float(len_div), _p(state['finished_count']), _p(state['best_score']),
        _p(state['best_len']), _p(state['best_tokens']), _p(state['new_input']),
        _p(state['parent']), _p(state['done']), _stream()), 'asr_beam_step_f32')


def _nhwc_bf16(t, what):
    """logical [B, C, H, W] bf16 tensor in channels-last memory -> (tensor, B, C, H, W)"""
    if not t.is_cuda:
        raise NativeLibraryError("%s must live on the MI355X" % what)
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    t = t.contiguous(memory_format=torch.channels_last)
    return (t,) + tuple(t.shape)


def conv7x7c32_fwd(x, weight, stride_h, want_sums=False):
    """asr_conv7x7c32_fwd_bf16: x logical [B, 32, H, W] (channels-last bf16), weight
    [32, 32, 7, 7] f32 -> y logical [B, 32, Ho, Wo] channels-last bf16 (and, want_sums, the
    [2, 32] f64 channel sums / sums of squares of y for the following BatchNorm)."""
    x, B, C, H, W = _nhwc_bf16(x, 'x')
    weight = _dev(weight, torch.float32, 'weight')
    Ho, Wo = (H - 7) // stride_h + 1, W - 6
    y = torch.empty((B, 32, Ho, Wo), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    L = lib()
    nbytes = L.asr_conv7x7c32_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    sums = torch.empty((2, 32), dtype=torch.float64, device=x.device) if want_sums else None
    check(L.asr_conv7x7c32_fwd_bf16(_p(x), _p(weight), B, H, W, int(stride_h), _p(y), _p(sums),
                                    _p(ws), nbytes, _stream()), 'asr_conv7x7c32_fwd_bf16')
    return (y, sums) if want_sums else y


def conv7x7c32_bwd_data(dy, weight, H, W, stride_h):
    """asr_conv7x7c32_bwd_data_bf16: dy logical [B, 32, Ho, Wo] -> dx logical [B, 32, H, W]
    (channels-last bf16)."""
    dy, B, C, Ho, Wo = _nhwc_bf16(dy, 'dy')
    weight = _dev(weight, torch.float32, 'weight')
    assert Ho == (H - 7) // stride_h + 1 and Wo == W - 6
    dx = torch.empty((B, 32, H, W), dtype=torch.bfloat16, device=dy.device,
                     memory_format=torch.channels_last)
    L = lib()
    nbytes = L.asr_conv7x7c32_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    check(L.asr_conv7x7c32_bwd_data_bf16(_p(dy), _p(weight), B, H, W, int(stride_h), _p(dx),
                                         _p(ws), nbytes, _stream()), 'asr_conv7x7c32_bwd_data_bf16')
    return dx


def conv7x7c32_wgrad(x, dy, stride_h):
    """asr_conv7x7c32_wgrad_bf16: x logical [B, 32, H, W], dy logical [B, 32, Ho, Wo]
    (channels-last bf16) -> dw [32, 32, 7, 7] f32."""
    x, B, C, H, W = _nhwc_bf16(x, 'x')
    dy = _nhwc_bf16(dy, 'dy')[0]
    dw = torch.empty((32, 32, 7, 7), dtype=torch.float32, device=x.device)
    L = lib()
    nbytes = L.asr_conv7x7c32_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(L.asr_conv7x7c32_wgrad_bf16(_p(x), _p(dy), B, H, W, int(stride_h), _p(dw), _p(ws),
                                      nbytes, _stream()), 'asr_conv7x7c32_wgrad_bf16')
    return dw


def conv1_supported(F, cin):
    """shapes csrc/conv.hip has a first convolution for: one channel (any width up to Fo = 64),
    or the WSJ recipes' 3 channels with an even output width"""
    Fo = (F - 7) // 2 + 1
    return F >= 7 and ((cin == 1 and Fo <= 64) or (cin == 3 and Fo <= 48 and Fo % 2 == 0))


def conv1_fwd(x, weight, want_sums=False):
    """asr_conv1_7x7s2_fwd / asr_conv1c_7x7s2_fwd: x [B, T, F] or [B, T, F, cin] f32, weight
    [32, cin, 7, 7] f32 -> y logical [B, 32, T + 6, (F - 7) // 2 + 1] channels-last bf16."""
    x = _dev(x, torch.float32, 'x')
    weight = _dev(weight, torch.float32, 'weight')
    cin = x.shape[3] if x.dim() == 4 else 1
    B, T, F = x.shape[:3]
    if tuple(weight.shape) != (32, cin, 7, 7):
        raise ValueError('conv1_fwd: weight must be [32, %d, 7, 7]' % cin)
    y = torch.empty((B, 32, T + 6, (F - 7) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    L = lib()
    sums = torch.empty((2, 32), dtype=torch.float64, device=x.device) if want_sums else None
    if cin == 1:
        nbytes = L.asr_conv1_7x7s2_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(L.asr_conv1_7x7s2_fwd(_p(x), _p(weight), B, T, F, _p(y), _p(sums), _p(ws), nbytes,
                                    _stream()), 'asr_conv1_7x7s2_fwd')
    else:
        nbytes = L.asr_conv1c_7x7s2_workspace_bytes(cin)
        if nbytes < 0:
            raise NotImplementedError('conv1_fwd: %d input channels' % cin)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(L.asr_conv1c_7x7s2_fwd(_p(x), _p(weight), B, T, F, cin, _p(y), _p(sums), _p(ws), nbytes,
                                     _stream()), 'asr_conv1c_7x7s2_fwd')
    return (y, sums) if want_sums else y


def conv1_wgrad(x, dy):
    """asr_conv1_7x7s2_wgrad / asr_conv1c_7x7s2_wgrad: x [B, T, F] or [B, T, F, cin] f32, dy
    logical [B, 32, To, Fo] channels-last bf16 -> dw [32, cin, 7, 7] f32."""
    x = _dev(x, torch.float32, 'x')
    dy = _nhwc_bf16(dy, 'dy')[0]
    cin = x.shape[3] if x.dim() == 4 else 1
    B, T, F = x.shape[:3]
    dw = torch.empty((32, cin, 7, 7), dtype=torch.float32, device=x.device)
    L = lib()
    if cin == 1:
        nbytes = L.asr_conv1_7x7s2_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(L.asr_conv1_7x7s2_wgrad(_p(x), _p(dy), B, T, F, _p(dw), _p(ws), nbytes, _stream()),
              'asr_conv1_7x7s2_wgrad')
    else:
        nbytes = L.asr_conv1c_7x7s2_workspace_bytes(cin)
        if nbytes < 0:
            raise NotImplementedError('conv1_wgrad: %d input channels' % cin)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(L.asr_conv1c_7x7s2_wgrad(_p(x), _p(dy), B, T, F, cin, _p(dw), _p(ws), nbytes, _stream()),
              'asr_conv1c_7x7s2_wgrad')
    return dw


def split_bf16(x, hi=None, lo=None):
    """asr_split_bf16_f32: x f32 on the GPU -> (hi, lo) bf16 with hi + lo = x to 2^-16.
    x contiguous of any shape (new contiguous outputs), or x [rows, cols] with given 2-D
    output views hi / lo whose last dimension is dense (the halves of a K-concatenated
    GEMM operand)."""
    x = _dev(x, torch.float32, 'x')
    if hi is None:
        hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        rows, cols, ldx, ldh, ldl = 1, x.numel(), x.numel(), x.numel(), x.numel()
    else:
        if x.dim() != 2 or hi.shape != x.shape or lo.shape != x.shape or x.stride(1) != 1 \
                or hi.stride(1) != 1 or lo.stride(1) != 1 or hi.dtype != torch.bfloat16 or lo.dtype != torch.bfloat16:
            raise ValueError('split_bf16: x, hi, lo must be [rows, cols] with a dense last dimension')
        rows, cols = x.shape
        ldx, ldh, ldl = x.stride(0), hi.stride(0), lo.stride(0)
    check(lib().asr_split_bf16_f32(_p(x), rows, cols, ldx, _p(hi), ldh, _p(lo), ldl, _stream()),
          'asr_split_bf16_f32')
    return hi, lo


def sum_leading(t):
    """asr_sum_leading_f32: t [G, ...] f32 contiguous on the GPU -> t.sum(0)"""
    if not t.is_cuda or t.dtype != torch.float32 or (t[0].numel() & 3):
        return t.sum(0)
    t = t.contiguous()
    out = torch.empty(t.shape[1:], dtype=torch.float32, device=t.device)
    check(lib().asr_sum_leading_f32(_p(t), t.shape[0], t[0].numel(), _p(out), _stream()),
          'asr_sum_leading_f32')
    return out


def log_softmax_shift_fwd(x, lens):
    """asr_log_softmax_shift_fwd_f32: x [T, B, C] -> (y = x - max_c x, nls [T, B], nls_sum [B])"""
    x = _dev(x, torch.float32, 'logits')
    lens = _dev(lens, torch.int32, 'lens')
    T, B, C = x.shape
    y = torch.empty_like(x)
    nls = torch.empty((T, B), dtype=torch.float32, device=x.device)
    nls_sum = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib().asr_log_softmax_shift_fwd_f32(_p(x), T, B, C, _p(lens), _p(y), _p(nls), _p(nls_sum),
                                              _stream()), 'asr_log_softmax_shift_fwd_f32')
    return y, nls, nls_sum


def log_softmax_shift_bwd_split(y, nls, dy, ld):
    """asr_log_softmax_shift_bwd_split_bf16: the gradient of log_softmax_shift_fwd as bf16
    halves (hi, lo) [rows, ld] (columns >= C zero) and its column sums [C]."""
    y, dy = _dev(y, torch.float32, 'y'), _dev(dy, torch.float32, 'dy')
    C = y.shape[-1]
    rows = y.numel() // C
    L = lib()
    hi = torch.empty((rows, ld), dtype=torch.bfloat16, device=y.device)
    lo = torch.empty((rows, ld), dtype=torch.bfloat16, device=y.device)
    part = torch.empty((L.asr_log_softmax_shift_bwd_split_blocks(rows), ld), dtype=torch.float32, device=y.device)
    check(L.asr_log_softmax_shift_bwd_split_bf16(_p(y), _p(nls), _p(dy), rows, C, _p(hi), _p(lo), int(ld),
                                                 _p(part), _stream()), 'asr_log_softmax_shift_bwd_split_bf16')
    nb = part.shape[0]
    if nb % 32 == 0 and nb > 32:         # two stages: sum_leading walks its leading axis serially
        part = sum_leading(part.view(32, (nb // 32) * ld)).view(nb // 32, ld)
    return hi, lo, sum_leading(part)[:C]


def log_softmax_shift_bwd(y, nls, dy):
    y, dy = _dev(y, torch.float32, 'y'), _dev(dy, torch.float32, 'dy')
    C = y.shape[-1]
    dx = torch.empty_like(y)
    check(lib().asr_log_softmax_shift_bwd_f32(_p(y), _p(nls), _p(dy), y.numel() // C, C, _p(dx),
                                              _stream()), 'asr_log_softmax_shift_bwd_f32')
    return dx
