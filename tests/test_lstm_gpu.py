"""Native BiLSTM recurrence (csrc/lstm.hip, bf16 MFMA operands / fp32 state)
against torch's fp32 nn.LSTM on a packed batch on the CPU — the op the
reference's BatchRNN runs (encoder_utils.py:78,100,112-117)."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(x, lens, rnn, dy):
    x = x.clone().requires_grad_()
    packed = nn.utils.rnn.pack_padded_sequence(x, lens)
    y, _ = rnn(packed)
    y, _ = nn.utils.rnn.pad_packed_sequence(y, total_length=x.size(0))
    y.backward(dy)
    return y.detach(), x.grad, [p.grad.clone() for p in rnn.parameters()]


def _bf(t):
    return t.to(torch.bfloat16).float()


def _emulate_bf16_operands(x, lens, rnn):
    """The arithmetic the kernels are built to do, spelled out with torch on the CPU: the
    operands of every matrix product (x, W_ih, h_{t-1}, W_hh) rounded to bf16, the products
    accumulated in fp32, gates / cell / output in fp32.  What separates the kernel from this
    is the order of the fp32 sums and a few ulp in exp / rcp — and, through the bf16 rounding
    of h, an occasional flipped last bit of an operand."""
    T, B, _ = x.shape
    H = rnn.hidden_size
    out = torch.zeros(T, B, 2, H)
    xb = _bf(x)
    for d, sfx in enumerate(('', '_reverse')):
        wih = _bf(getattr(rnn, 'weight_ih_l0' + sfx).detach())
        whh = _bf(getattr(rnn, 'weight_hh_l0' + sfx).detach())
        for b in range(B):
            L = int(lens[b])
            h = torch.zeros(H)
            c = torch.zeros(H)
            for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
                g = wih @ xb[t, b] + whh @ _bf(h)
                i, f, gg, o = g[:H].sigmoid(), g[H:2 * H].sigmoid(), g[2 * H:3 * H].tanh(), g[3 * H:].sigmoid()
                c = f * c + i * gg
                h = o * c.tanh()
                out[t, b, d] = h
    return out.view(T, B, 2 * H)


@pytest.mark.parametrize('T,B,F,H', [(37, 5, 48, 64), (40, 24, 352, 320), (25, 33, 320, 320)])
def test_bilstm_is_the_bf16_operand_evaluation(T, B, F, H, monkeypatch):
    """Whose error is the 3e-2 of the comparison with fp32 below?  The rounding of the
    operands to bf16, not the kernels: against the bf16-operand evaluation above the outputs
    agree to 2e-3 of their range (15 times closer than to the fp32 model), and that
    evaluation itself sits as far from fp32 as the kernels do."""
    from att_speech.modules.encoders.native_lstm import bilstm
    monkeypatch.setenv('ASR_GX_FP32', '1')          # (paths that run x.W_ih as a GEMM keep it in fp32)
    torch.manual_seed(T * 77 + B)
    lens = sorted(np.random.RandomState(B + 1).randint(1, T + 1, size=B).tolist(), reverse=True)
    lens[0] = T
    lens_t = torch.tensor(lens)
    rnn = nn.LSTM(F, H, bidirectional=True, bias=False)
    x = torch.randn(T, B, F)
    with torch.no_grad():
        emu = _emulate_bf16_operands(x, lens_t, rnn)
        packed = nn.utils.rnn.pack_padded_sequence(x, lens_t)
        y32, _ = nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], total_length=T)
    dev = torch.device('cuda:0')
    rnn_g = nn.LSTM(F, H, bidirectional=True, bias=False)
    rnn_g.load_state_dict(rnn.state_dict())
    rnn_g.to(dev)
    with torch.no_grad():
        y = bilstm(x.to(dev), lens_t, rnn_g).view(T, B, 2 * H).cpu()
    scale = float(y32.abs().max())
    to_emu = float((y - emu).abs().max())
    emu_to_32 = float((emu - y32).abs().max())
    gpu_to_32 = float((y - y32).abs().max())
    assert to_emu <= 2e-3 * scale, (to_emu, scale)
    assert float((y - emu).abs().mean()) <= 5e-5 * scale
    assert gpu_to_32 <= 1.5 * emu_to_32 + 2e-3 * scale, (gpu_to_32, emu_to_32)


@pytest.mark.parametrize('T,B,F,H,lens', [
    (37, 5, 48, 64, [37, 30, 30, 11, 1]),
    (60, 40, 352, 320, None),          # first encoder layer shape, 2 batch tiles
    (12, 33, 320, 320, None),
])
@pytest.mark.parametrize('sum_dirs', [False, True])
def test_bilstm_matches_packed_torch_lstm(T, B, F, H, lens, sum_dirs):
    from att_speech.modules.encoders.native_lstm import bilstm
    torch.manual_seed(T * 1000 + B)
    if lens is None:
        lens = sorted(np.random.RandomState(B).randint(1, T + 1, size=B).tolist(), reverse=True)
        lens[0] = T
    lens_t = torch.tensor(lens)
    rnn = nn.LSTM(F, H, bidirectional=True, bias=False)
    x = torch.randn(T, B, F)
    dy = torch.randn(T, B, 2 * H)
    mask = (torch.arange(T)[:, None] < lens_t[None, :]).float()[:, :, None]
    dy = dy * mask
    if sum_dirs:        # BatchRNN's merge: the two directions share one gradient
        g = dy.view(T, B, 2, H)[:, :, 0].contiguous()
        dy = torch.stack([g, g], 2).view(T, B, 2 * H)
    y_ref, dx_ref, dw_ref = _ref(x, lens_t, rnn, dy)

    dev = torch.device('cuda:0')
    rnn_g = nn.LSTM(F, H, bidirectional=True, bias=False)
    rnn_g.load_state_dict(rnn.state_dict())
    rnn_g.to(dev)
    xg = x.to(dev).requires_grad_()
    if sum_dirs:
        y = bilstm(xg, lens_t, rnn_g, sum_dirs=True)
        assert tuple(y.shape) == (T, B, H)
        y.backward(g.to(dev))
        y_ref = y_ref.view(T, B, 2, H).sum(2)
    else:
        y = bilstm(xg, lens_t, rnn_g).view(T, B, 2 * H)
        y.backward(dy.to(dev))

    def close(a, b, what, rtol):
        a, b = a.detach().cpu(), b.detach().cpu()
        err = float((a - b).abs().max())
        scale = float(b.abs().max()) + 1e-6
        assert err <= rtol * scale, (what, err, scale)

    # bf16 operands (8-bit mantissa), fp32 accumulation and state
    close(y, y_ref, 'y', 3e-2)
    assert not y.detach().cpu()[mask.expand_as(y_ref) == 0].any()   # zeros on padding
    close(xg.grad, dx_ref, 'dx', 5e-2)
    for p, w in zip(rnn_g.parameters(), dw_ref):
        close(p.grad, w, 'dw', 5e-2)


def _run_native(gx, whh, lens, dy, persist):
    import os
    from att_speech import _native
    os.environ['ASR_LSTM_PERSIST'] = '1' if persist else '0'
    try:
        y, ybf, gates, csave = _native.lstm_bidir_fwd(gx, whh, lens)
        whhT = whh.view(2, -1, whh.size(-1)).transpose(1, 2).contiguous()
        dg = _native.lstm_bidir_bwd(dy, whhT, lens, gates, csave)
        torch.cuda.synchronize()
    finally:
        os.environ.pop('ASR_LSTM_PERSIST', None)
    return y, ybf, gates, csave, dg


@pytest.mark.parametrize('T,B,H,reps', [
    (1, 1, 64, 1),             # a single frame, a single utterance
    (2, 3, 64, 1),
    (23, 7, 64, 1),            # one workgroup per team, ragged batch tile
    (61, 45, 128, 1),          # two-workgroup teams
    (150, 96, 320, 1),         # five-workgroup teams (the encoder's hidden size)
    (334, 512, 320, 3),        # bench shape: 160 workgroups, every hand-off cross-XCD
    (40, 900, 320, 1),         # more batch tiles than CUs allow in one launch
    (30, 64, 512, 1),          # backward falls back to one launch per step (LDS)
])
def test_persistent_recurrence_is_bitwise_the_per_step_one(T, B, H, reps):
    """The persistent kernels (teams handing h_t / dgates_t over through L2 inside
    one launch) do the same arithmetic in the same order as the one-launch-per-
    step kernels, so every output must agree bit for bit; one stale or torn
    hand-off changes bits.  Repeated at the bench shape with uneven lengths."""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(T * 31 + B)
    lens = torch.randint(1, T + 1, (B,), generator=g).sort(descending=True)[0]
    lens[0] = T
    gx = (torch.randn(T, B, 2, 4 * H, generator=g) * 1.5).to(dev)
    whh = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev, torch.bfloat16)
    dy = torch.randn(T, B, 2, H, generator=g).to(dev)
    lens_d = lens.to(dev, torch.int32)
    act = (torch.arange(T)[:, None] < lens[None, :]).to(dev)           # [T,B]
    runs = [gx] * reps + ([gx.to(torch.bfloat16)] if reps == 1 else [])   # fp32 and bf16 x.W_ih
    ref_cache = {}
    for gxi in runs:
        if gxi.dtype not in ref_cache:
            ref_cache[gxi.dtype] = _run_native(gxi, whh, lens_d, dy, persist=False)
        ref = ref_cache[gxi.dtype]
        out = _run_native(gxi, whh, lens_d, dy, persist=True)
        for name, a, b in zip(('y', 'y_bf16', 'gates', 'csave', 'dgates'), out, ref):
            if name == 'gates':                                         # [T,2,B,H,4]: defined on active frames
                m = act[:, None, :, None, None].expand_as(a)
                a, b = a[m], b[m]
            assert not torch.isnan(a.float()).any(), name
            assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize('T,B,H,reps', [
    (1, 1, 64, 1),
    (23, 7, 64, 1),            # one workgroup per team, ragged batch tile
    (61, 45, 128, 1),
    (19, 70, 256, 1),
    (150, 96, 320, 1),         # 16-row batch tiles
    (334, 512, 320, 3),        # bench shape (24-row tiles), repeated
    (40, 900, 320, 1),         # 32-row tiles in several launches
])
def test_fused_input_projection_matches_gemm_plus_recurrence(T, B, H, reps):
    _fused_projection_case(T, B, H, H, reps)


@pytest.mark.parametrize('T,B,reps', [(1, 1, 1), (23, 7, 1), (150, 96, 1), (334, 576, 2), (40, 900, 1)])
def test_fused_input_projection_first_layer_shape(T, B, reps):
    """the 352-feature input of the first layer behind the conv front-end (22 k-steps; that
    variant writes the bf16 outputs only)"""
    _fused_projection_case(T, B, 320, 352, reps)


@pytest.mark.parametrize('T,B,F', [(2, 3, 320), (23, 7, 320), (24, 40, 320), (151, 96, 320), (334, 576, 320),
                                   (334, 768, 320), (40, 900, 320), (23, 7, 352), (150, 96, 352), (334, 576, 352)])
def test_direction_sum_inside_the_recurrence(T, B, F):
    """asr_lstm_bidir_fwd_fused_sum_bf16: the recurrence as two launches (steps [0, ceil(T/2)) and
    the rest, the second picking its state up from the hand-off buffer / csave), the second
    adding the other direction's bf16 output of each frame to its own.  Every output must be
    bit-identical to the single launch (same arithmetic; a wrong restored state, a stale tile
    or a missed frame changes bits), xsum bit-identical to the add over the two planes; odd
    and even T, ragged lengths, 16/24/32-row tiles, a batch that needs several launches."""
    from att_speech import _native
    dev = torch.device('cuda:0')
    H = 320
    if not _native.experiments_built():
        pytest.skip('experiment: needs a `make EXPERIMENTS=1` library (include/asr_amd_experiments.h)')
    if not _native.lstm_fused_supported(B, H, F=F, dirsum=True):
        pytest.skip('no direction-sum kernel for this (batch, input size)')
    g = torch.Generator().manual_seed(T * 13 + B + F)
    lens = torch.randint(1, T + 1, (B,), generator=g).sort(descending=True)[0]
    lens[0] = T
    x = torch.randn(T, B, F, generator=g).to(dev, torch.bfloat16)
    wih = (torch.randn(8 * H, F, generator=g) * (1.0 / F ** 0.5)).to(dev, torch.bfloat16)
    whh = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev, torch.bfloat16)
    lens_d = lens.to(dev, torch.int32)
    want_y = F == H
    one = _native.lstm_bidir_fwd_fused(x, wih, whh, lens_d, want_y=want_y)
    act = (torch.arange(T)[:, None] < lens[None, :]).to(dev)
    for _ in range(2):
        two = _native.lstm_bidir_fwd_fused(x, wih, whh, lens_d, want_y=want_y, want_sum=True)
        torch.cuda.synchronize()
        _native.lstm_check_errors()
        for name, a, b in zip(('y', 'y_bf16', 'gates', 'csave'), two, one):
            if a is None:
                continue
            if name == 'gates':         # records of padding frames are never written
                m = act[:, None, :, None, None].expand_as(a)
                a, b = a[m], b[m]
            assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32),
                               b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)), name
        want = one[1][0, 1:T + 1] + one[1][1, 1:T + 1]
        assert torch.equal(two[4].view(torch.int16), want.view(torch.int16))


def _fused_projection_case(T, B, H, F, reps):
    """asr_lstm_bidir_fwd_fused_bf16 (x_t·W_ih inside the persistent kernel, the tile brought
    in by LDS-DMA under the hand-off waits) against the fp32 product of the same bf16
    operands fed to asr_lstm_bidir_fwd_bf16.  Same arithmetic up to the order of the fp32
    accumulation (the x term opens the MFMA chain instead of being added after it).  A
    last-bit difference in a pre-activation can flip the bf16 rounding of an h_t that feeds
    the next step, so over hundreds of steps the two runs drift by a few bf16 ulps of single
    elements (measured 7e-4 at T=150; a stale tile would be O(1)): 5e-3 on the fp32 outputs,
    2e-2 on the bf16 records; every repetition of the fused kernel is bit-identical to the first (a stale x tile
    or a torn hand-off would change bits)."""
    from att_speech import _native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(T * 17 + B)
    lens = torch.randint(1, T + 1, (B,), generator=g).sort(descending=True)[0]
    lens[0] = T
    x = torch.randn(T, B, F, generator=g).to(dev, torch.bfloat16)
    wih = (torch.randn(8 * H, F, generator=g) * (1.0 / F ** 0.5)).to(dev, torch.bfloat16)
    whh = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev, torch.bfloat16)
    lens_d = lens.to(dev, torch.int32)
    assert _native.lstm_fused_supported(B, H, F=F)
    want_y = F == H
    gx = torch.mm(x.view(T * B, F), wih.t(), out_dtype=torch.float32).view(T, B, 2, 4 * H)
    ref = _native.lstm_bidir_fwd(gx, whh, lens_d)
    act = (torch.arange(T)[:, None] < lens[None, :]).to(dev)
    first = None
    for _ in range(reps):
        out = _native.lstm_bidir_fwd_fused(x, wih, whh, lens_d, want_y=want_y)
        torch.cuda.synchronize()
        _native.lstm_check_errors()
        if first is None:
            first = out
            for name, a, b in zip(('y', 'y_bf16', 'gates', 'csave'), out, ref):
                if a is None:
                    continue
                if name == 'gates':
                    m = act[:, None, :, None, None].expand_as(a)
                    a, b = a[m], b[m]
                a, b = a.float(), b.float()
                assert not torch.isnan(a).any(), name
                tol = 2e-2 if name in ('y_bf16', 'gates') else 5e-3
                err = float((a - b).abs().max()) if a.numel() else 0.0
                assert err <= tol * (1.0 + float(b.abs().max()) if b.numel() else 1.0), (name, err)
        else:
            for name, a, b in zip(('y', 'y_bf16', 'gates', 'csave'), out, first):
                if a is None:
                    continue
                if name == 'gates':
                    m = act[:, None, :, None, None].expand_as(a)
                    a, b = a[m], b[m]
                assert torch.equal(a, b), name
    # y == None: the inner layers of a stack ask for the bf16 plane only
    out = _native.lstm_bidir_fwd_fused(x, wih, whh, lens_d, want_y=False)
    assert out[0] is None and torch.equal(out[1], first[1])


@pytest.mark.parametrize('T,B,H,reps', [
    (1, 1, 64, 1),
    (2, 3, 64, 1),
    (23, 7, 64, 1),
    (61, 45, 128, 1),
    (19, 70, 256, 1),
    (150, 96, 320, 1),
    (334, 512, 320, 3),        # bench shape, repeated
])
def test_fused_input_gradient_matches_recurrence_plus_gemm(T, B, H, reps):
    """asr_lstm_bidir_bwd_fused_bf16 (dx = dgates·W_ih inside the persistent backward kernel,
    hand-off tile and pointwise operands by LDS-DMA) against asr_lstm_bidir_bwd_bf16 + the
    GEMM: the gate gradients go through the same arithmetic in the same order, so they must
    agree bit for bit (a stale tile, a late operand or a torn hand-off changes bits); the
    sum of the two dx planes is the fp32 product of the same bf16 operands up to summation
    order.  Also: a gradient handed in as two planes equals the same gradient pre-summed."""
    from att_speech import _native
    dev = torch.device('cuda:0')
    if not _native.experiments_built():
        pytest.skip('experiment: needs a `make EXPERIMENTS=1` library (include/asr_amd_experiments.h)')
    g = torch.Generator().manual_seed(T * 13 + B)
    lens = torch.randint(1, T + 1, (B,), generator=g).sort(descending=True)[0]
    lens[0] = T
    gx = (torch.randn(T, B, 2, 4 * H, generator=g) * 1.5).to(dev)
    whh = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev, torch.bfloat16)
    wih = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev, torch.bfloat16)
    dya = torch.randn(T, B, H, generator=g).to(dev)
    dyb = torch.randn(T, B, H, generator=g).to(dev)
    lens_d = lens.to(dev, torch.int32)
    assert _native.lstm_fused_supported(B, H, backward=True)
    _, _, gates, csave = _native.lstm_bidir_fwd(gx, whh, lens_d)
    whhT, wihT = whh.transpose(1, 2).contiguous(), wih.transpose(1, 2).contiguous()
    dy = dya + dyb
    want = _native.lstm_bidir_bwd(dy, whhT, lens_d, gates, csave)
    want_dx = torch.mm(want.view(T * B, 8 * H), wih.view(8 * H, H), out_dtype=torch.float32).view(T, B, H)
    for _ in range(reps):
        for planes in (False, True):
            arg = torch.stack([dya, dyb]) if planes else dy
            dg, dx = _native.lstm_bidir_bwd_fused(arg, whhT, wihT, lens_d, gates, csave, planes=planes)
            torch.cuda.synchronize()
            _native.lstm_check_errors()
            assert torch.equal(dg, want), float((dg.float() - want.float()).abs().max())
            got = dx[0] + dx[1]
            assert float((got - want_dx).abs().max()) <= 2e-3 * (1.0 + float(want_dx.abs().max()))
    # the unfused kernel takes planes too (the layer below a fused one)
    dg = _native.lstm_bidir_bwd(torch.stack([dya, dyb]), whhT, lens_d, gates, csave, planes=True)
    assert torch.equal(dg, want)


def test_handoff_timeout_surfaces_as_an_error(monkeypatch):
    """A persistent-recurrence hand-off whose spin bound expires poisons the outputs with
    NaN and sets the caller's error word; `_native.lstm_check_errors` (called once per step
    by dp.train_step) must raise instead of letting the NaNs reach the weights.  The
    timeout is forced with the debug bound ASR_LSTM_SPIN_LIMIT=0; a normal call afterwards
    is clean again."""
    from att_speech import _native
    from att_speech.modules.encoders.native_lstm import bilstm
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    T, B, F, H = 6, 40, 64, 320          # H = 320: five workgroups per team -> real hand-offs
    rnn = nn.LSTM(F, H, bidirectional=True, bias=False).to(dev)
    x = torch.randn(T, B, F, device=dev)
    lens = torch.full((B,), T, dtype=torch.int64)
    _native.lstm_check_errors()                       # clean before
    monkeypatch.setenv('ASR_LSTM_SPIN_LIMIT', '0')
    y = bilstm(x, lens, rnn)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='hand-off timed out'):
        _native.lstm_check_errors()
    assert not bool(torch.isfinite(y).all())
    monkeypatch.delenv('ASR_LSTM_SPIN_LIMIT')
    y = bilstm(x, lens, rnn)
    _native.lstm_check_errors()                       # the word was cleared by the raise
    assert bool(torch.isfinite(y).all())


@pytest.mark.parametrize('fused_bwd', [False, True])
def test_layer_stack_matches_chained_layers(fused_bwd, monkeypatch):
    """native_lstm.bilstm_stack (one autograd node, bf16 planes between layers, no fp32 outputs
    of the inner layers) against the same layers applied one by one with summed directions:
    outputs and all gradients agree at the level of the bf16 inter-layer operand."""
    from att_speech.modules.encoders.native_lstm import bilstm, bilstm_stack
    if fused_bwd:       # the opt-in input-gradient fusion (dx planes handed from layer to layer)
        monkeypatch.setenv('ASR_LSTM_FUSED_BWD', '1')
    torch.manual_seed(5)
    dev = torch.device('cuda:0')
    T, B, F, H = 23, 37, 96, 128
    lens = torch.tensor(sorted(np.random.RandomState(3).randint(1, T + 1, size=B).tolist(), reverse=True))
    lens[0] = T
    rnns = [nn.LSTM(F if l == 0 else H, H, bidirectional=True, bias=False).to(dev) for l in range(3)]
    x = torch.randn(T, B, F, device=dev)
    dy = torch.randn(T, B, H, device=dev) * (torch.arange(T, device=dev)[:, None, None] < lens.to(dev)[None, :, None])

    xa = x.clone().requires_grad_()
    y = xa
    for r in rnns:
        y = bilstm(y, lens, r, sum_dirs=True)
    y.backward(dy)
    want = [y.detach(), xa.grad] + [p.grad.clone() for r in rnns for p in r.parameters()]
    for r in rnns:
        r.zero_grad()
    xb = x.clone().requires_grad_()
    y2 = bilstm_stack(xb, lens, rnns)
    y2.backward(dy)
    got = [y2.detach(), xb.grad] + [p.grad for r in rnns for p in r.parameters()]
    for a, b_ in zip(got, want):
        scale = float(b_.abs().max())
        assert float((a - b_).abs().max()) <= 2e-2 * scale + 1e-6


@pytest.mark.parametrize('T,B,with_input', [
    (1, 1, True),
    (7, 5, True),              # fewer frames than one stage per chunk
    (50, 33, True),
    (50, 33, False),           # first-layer form: dW_hh only
    (334, 96, True),
    (334, 576, True),          # bench shape
    (334, 576, False),
])
def test_weight_gradient_kernel_matches_fp32_products(T, B, with_input):
    """asr_lstm_wgrad_bf16 (one pass over the gate gradients: LDS-DMA ring, transposing LDS
    reads, MFMA) against the fp32 products of the same bf16 operands in torch, with exact
    integer-valued data first (any mis-indexed fragment, swizzle or tail frame shows up as an
    exact mismatch) and random data second (fp32 summation order differs: 2e-3 of the
    largest entry)."""
    from att_speech import _native
    dev = torch.device('cuda:0')
    H = 320
    assert _native.lstm_wgrad_supported(H)
    g = torch.Generator().manual_seed(T * 7 + B)
    for exact in (True, False):
        if exact:       # small integers: every product and partial sum is exact in fp32
            dg = torch.randint(-2, 3, (T, B, 2, 4 * H), generator=g).float()
            x = torch.randint(-2, 3, (T * B, H), generator=g).float()
            y = torch.randint(-2, 3, (2, T + 2, B, H), generator=g).float()
            if T * B > 4096:          # keep |sums| < 2^24
                dg = dg * (torch.rand(T, B, 1, 1, generator=g) < 4096.0 / (T * B)).float()
        else:
            dg = torch.randn(T, B, 2, 4 * H, generator=g)
            x = torch.randn(T * B, H, generator=g)
            y = torch.randn(2, T + 2, B, H, generator=g)
        dg, x, y = dg.to(dev, torch.bfloat16), x.to(dev, torch.bfloat16), y.to(dev, torch.bfloat16)
        dw_ih, dw_hh = _native.lstm_wgrad(dg, x if with_input else None, y)
        torch.cuda.synchronize()
        a = dg.view(T * B, 8 * H).float()
        want_hh = torch.stack([
            a[:, :4 * H].t() @ y[0, 0:T].reshape(T * B, H).float(),
            a[:, 4 * H:].t() @ y[1, 2:T + 2].reshape(T * B, H).float()])
        pairs = [(dw_hh, want_hh)]
        if with_input:
            pairs.append((dw_ih, a.t() @ x.float()))
        else:
            assert dw_ih is None
        for got, want in pairs:
            if exact:
                assert torch.equal(got, want), float((got - want).abs().max())
            else:
                assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max())


@pytest.mark.parametrize('T,B', [(1, 1), (7, 5), (50, 33), (334, 96), (334, 576)])
def test_input_gradient_kernel_matches_fp32_product(T, B):
    """asr_lstm_dgrad_bf16 (dx = dgates · W_ih: LDS-DMA stages, swizzled row-major A through
    ds_read_b128, k-major W_ih through transposing reads, MFMA) against the fp32 product of the
    same bf16 operands: exact on small-integer data, 2e-3 of the largest entry on random data."""
    from att_speech import _native
    dev = torch.device('cuda:0')
    H = 320
    assert _native.lstm_dgrad_supported(H)
    g = torch.Generator().manual_seed(T * 5 + B)
    for exact in (True, False):
        if exact:
            dg = torch.randint(-2, 3, (T, B, 2, 4 * H), generator=g).float()
            w = torch.randint(-2, 3, (8 * H, H), generator=g).float()
        else:
            dg = torch.randn(T, B, 2, 4 * H, generator=g)
            w = torch.randn(8 * H, H, generator=g) * 0.05
        dg, w = dg.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16)
        dx = _native.lstm_dgrad(dg, w)
        torch.cuda.synchronize()
        want = (dg.view(T * B, 8 * H).float() @ w.float()).view(T, B, H)
        if exact:
            assert torch.equal(dx, want), float((dx - want).abs().max())
        else:
            assert float((dx - want).abs().max()) <= 2e-3 * float(want.abs().max())


def test_long_batches_leave_the_fused_projection(monkeypatch):
    """A batch whose T*B exceeds the fused kernel's 32-bit offsets (B=768 with T' >= 547 at
    H=320) must take the GEMM + recurrence path instead of failing with ASR_EUNSUPPORTED
    (ADVICE r2).  The limit is lowered here instead of allocating a 4 GiB tensor; both paths
    give the same layer output up to the bf16 rounding of the projected input."""
    from att_speech import _native
    from att_speech.modules.encoders import native_lstm
    dev = torch.device('cuda:0')
    T, B, H = 40, 48, 320
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, H, generator=g).to(dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    w = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev) for _ in range(4)]
    calls = {'fused': 0, 'plain': 0}
    real_fused, real_plain = _native.lstm_bidir_fwd_fused, _native.lstm_bidir_fwd
    monkeypatch.setattr(_native, 'lstm_bidir_fwd_fused',
                        lambda *a, **k: (calls.__setitem__('fused', calls['fused'] + 1), real_fused(*a, **k))[1])
    monkeypatch.setattr(_native, 'lstm_bidir_fwd',
                        lambda *a, **k: (calls.__setitem__('plain', calls['plain'] + 1), real_plain(*a, **k))[1])
    y1 = native_lstm.BiLSTMFunction.apply(x, lens, w[0], w[1], w[2], w[3], True)
    assert calls == {'fused': 1, 'plain': 0}
    monkeypatch.setitem(native_lstm._FUSED_LIMITS, 'gates', T * B * 8 * H * 4)      # "does not fit"
    y2 = native_lstm.BiLSTMFunction.apply(x, lens, w[0], w[1], w[2], w[3], True)
    assert calls == {'fused': 1, 'plain': 1}
    torch.cuda.synchronize()
    _native.lstm_check_errors()
    assert float((y1 - y2).abs().max()) < 3e-2 * float(y1.abs().max())
