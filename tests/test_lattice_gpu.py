"""GPU parity tests proper: the HIP path, called through the C ABI
(att_speech._native -> libasr_amd.so), against (a) the committed golden vectors
of the imported reference, (b) the CPU oracle on seeded inputs, (c) size-free
properties at the benchmark's full sizes.

Tolerances (north_star): loss within 1e-4 relative fp32; best-path label
indices bit-exact."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

LATTICES = ['lattice_mono', 'lattice_bigram_s7', 'lattice_bigram_s49',
            'lattice_den_mono', 'lattice_den_bigram_s7']
RTOL_LOSS = 1e-4
ATOL_GRAD = 2e-5          # occupancies are in [0,1]


def grad_atol(logZ):
    """Occupancies are exp(alpha + beta - logZ) with |alpha|, |beta| ~ |logZ|, so
    their fp32 conditioning scales with the magnitude of the scores: the
    reference's own forward and backward totals differ by this much
    (fst_utils.py:475-479 tolerates 1e-3)."""
    return max(ATOL_GRAD, 4 * np.finfo(np.float32).eps * float(np.abs(logZ).max()))


def assert_posteriors(grad, ref32, lp, lens, mats, oracle_lib):
    """Posterior (gradient) parity, arbitrated in fp64.

    `ref32` is an fp32 evaluation of the reference arithmetic (the C oracle, or the imported
    reference's own output from a golden file).  Both it and the kernel round
    exp(alpha + beta - logZ) in fp32, and on long lattices the two roundings differ by more
    than a flat 2e-5 (history in DESIGN.md §2: the bound was widened twice after red runs).
    Instead of a wider bound, the fp64 evaluation of the same recurrences
    (oracle_path_logsumexp_f64) says whose error it is: the kernel may be off from fp64 by at
    most max(2e-5, 2 x the fp32 reference's own distance from fp64) in the max norm, and by
    max(2e-6, 2 x the reference's) in the mean, so a systematic bias cannot hide inside the
    max-abs bound.  Utterances with no feasible alignment (logZ ~ -1e20) are left out: their
    "posteriors" are exp() of differences of +-1e20 sentinels (ulp 9e12), artefacts of the
    summation order that the reference neither uses nor defines; their logZ is compared by
    the callers, and until round 3 the bound 4 eps |logZ| silently made this whole comparison
    vacuous whenever a batch contained one."""
    f64 = oracle_lib.path_logsumexp_f64(lp, lens, mats)
    feas = f64['logZ'] > -1e19
    assert np.isfinite(grad).all()
    if not feas.any():
        return
    g, r, d = grad[:, feas].astype(np.float64), ref32[:, feas].astype(np.float64), f64['grad'][:, feas]
    err_k, err_r = np.abs(g - d), np.abs(r - d)
    assert err_k.max() <= max(ATOL_GRAD, 2 * err_r.max()), (err_k.max(), err_r.max())
    assert err_k.mean() <= max(ATOL_GRAD / 10, 2 * err_r.mean()), (err_k.mean(), err_r.mean())


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device('cuda:0')


def to_t(mats):
    return [torch.from_numpy(np.ascontiguousarray(m)) for m in mats]


def run_fwbw(lp, lens, mats, want_bwd=False, band=None):
    from att_speech import _native
    d = dev()
    g = _native.Graph(to_t(mats), d)
    if band is not None:
        g.band = band
        if band:            # the kernel under test, whatever the redo-rate policy decided earlier
            _native._BAND_STATE.update(cool=0, pending=None)
            _native._BAND_MAX_BATCH = 1 << 30
    logZ, grad, zb = _native.lattice_fwbw(
        torch.from_numpy(lp).to(d), torch.from_numpy(np.asarray(lens, np.int32)).to(d),
        g, -1e20, want_bwd_total=want_bwd)
    torch.cuda.synchronize()
    return logZ.cpu().numpy(), grad.cpu().numpy(), (zb.cpu().numpy() if want_bwd else None)


def run_fwd(lp, lens, mats, viterbi):
    from att_speech import _native
    d = dev()
    g = _native.Graph(to_t(mats[:4]), d)
    s, best = _native.lattice_forward(
        torch.from_numpy(lp).to(d), torch.from_numpy(np.asarray(lens, np.int32)).to(d),
        g, -1e20, viterbi=viterbi, want_path=viterbi)
    torch.cuda.synchronize()
    return s.cpu().numpy(), (best.cpu().numpy() if best is not None else None)


@pytest.mark.parametrize('name', LATTICES)
def test_golden_fwbw(oracle_lib, name):
    g = golden(name + '.npz')
    mats = [g['gm%d' % i] for i in range(8)]
    logZ, grad, zb = run_fwbw(g['lp'], g['lens'], mats, want_bwd=True)
    np.testing.assert_allclose(logZ, g['fwbw_logZ'], rtol=RTOL_LOSS)
    np.testing.assert_allclose(grad, g['fwbw_grad'], atol=grad_atol(logZ))
    assert_posteriors(grad, g['fwbw_grad'], g['lp'], g['lens'], mats, oracle_lib)
    assert np.abs(zb - logZ).max() < 1e-3                # fst_utils.py:475-479
    for b, l in enumerate(g['lens']):
        assert not grad[l:, b].any()                     # fst_utils.py:448


@pytest.mark.parametrize('name', LATTICES)
def test_golden_forward_and_viterbi(name):
    g = golden(name + '.npz')
    mats = [g['gm%d' % i] for i in range(8)]
    s, _ = run_fwd(g['lp'], g['lens'], mats, viterbi=False)
    np.testing.assert_allclose(s, g['autodiff_logZ'], rtol=RTOL_LOSS)
    v, best = run_fwd(g['lp'], g['lens'], mats, viterbi=True)
    np.testing.assert_allclose(v, g['viterbi_score'], rtol=1e-6)
    for b, l in enumerate(g['lens']):                    # bit-exact indices
        np.testing.assert_array_equal(best[:l, b], g['viterbi_selidx'][:l, b])
        assert not best[l:, b].any()


def _random_case(order, S, T, B, Lmax, seed, kind='num', lens=None):
    from att_speech import fst_utils as P
    rng = np.random.default_rng(seed)
    C = S ** order
    if lens is None:
        lens = np.sort(rng.integers(max(1, T // 2), T + 1, size=B))[::-1].copy()
        lens[0] = T
    lens = np.asarray(lens, np.int32)
    x = rng.standard_normal((T, B, C)).astype(np.float32) * 2
    pg = P.CTCGraphGen(context_order=order, num_symbols=S)
    if kind == 'num':
        lp = torch.log_softmax(torch.from_numpy(x), -1).numpy()
        llens = rng.integers(0, Lmax + 1, size=B)
        llens[0] = Lmax
        labs = rng.integers(1, S, size=(B, Lmax))
        mats = [m.numpy() for m in pg.get_training_matrices_batch(labs, llens)]
    else:
        lp = x - x.max(-1, keepdims=True)
        mats = [m.numpy() for m in pg.get_decoding_matrices()]
    return lp, lens, mats


CASES = [
    ('mono_num', dict(order=1, S=49, T=120, B=9, Lmax=40, seed=1)),
    ('mono_num_long', dict(order=1, S=49, T=700, B=3, Lmax=330, seed=2)),   # N=661 > 512
    ('bi_num', dict(order=2, S=49, T=60, B=5, Lmax=20, seed=3)),
    ('mono_den', dict(order=1, S=49, T=50, B=4, Lmax=0, seed=4, kind='den')),
    ('bi_den_s7', dict(order=2, S=7, T=50, B=4, Lmax=0, seed=5, kind='den')),
    ('bi_den_s49', dict(order=2, S=49, T=12, B=2, Lmax=0, seed=6, kind='den')),  # 2401 x 51
    ('ragged', dict(order=1, S=49, T=30, B=6, Lmax=8, seed=7,
                    lens=[30, 30, 17, 2, 1, 0])),
    ('single', dict(order=1, S=49, T=1, B=1, Lmax=1, seed=8, lens=[1])),
]


@pytest.mark.parametrize('name,kw', CASES, ids=[c[0] for c in CASES])
def test_seeded_vs_oracle(oracle_lib, name, kw):
    lp, lens, mats = _random_case(**kw)
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    logZ, grad, zb = run_fwbw(lp, lens, mats, want_bwd=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    assert_posteriors(grad, want['grad'], lp, lens, mats, oracle_lib)
    np.testing.assert_allclose(zb, want['logZ_bwd'], rtol=RTOL_LOSS, atol=1e-4)
    s, _ = run_fwd(lp, lens, mats, viterbi=False)
    np.testing.assert_allclose(s, want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    vs, vil = oracle_lib.path_forward(lp, lens, mats, viterbi=True)
    v, best = run_fwd(lp, lens, mats, viterbi=True)
    np.testing.assert_allclose(v, vs, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(best, vil)            # bit-exact, incl. padding rows


def test_generic_graph_with_per_arc_labels(oracle_lib):
    """A graph that is NOT state-labelled (in-arcs of a state carry different
    input labels) takes the generic body of the same launch; mixed batches
    (some utterances state-labelled, some not) are handled per workgroup."""
    from att_speech import fst_utils as P
    rng = np.random.default_rng(21)
    C, T, B, N = 11, 45, 4, 37
    per_utt = []
    for b in range(B):
        src, dst, il = [], [], []
        for n in range(N):
            for k in range(3):                       # <= 3 in-arcs per state
                s_ = int(rng.integers(max(0, n - 4), n + 1))
                if (s_, n) in zip(src, dst):
                    continue
                src.append(s_); dst.append(n)
                # utterance 0 stays state-labelled, the others do not
                il.append(n % C if b == 0 else int(rng.integers(0, C)))
        # at most 4 out-arcs per state (drop the excess)
        keep, outdeg = [], {}
        for i, s_ in enumerate(src):
            if outdeg.get(s_, 0) < 4:
                outdeg[s_] = outdeg.get(s_, 0) + 1
                keep.append(i)
        src, dst, il = [np.array(x)[keep] for x in (src, dst, il)]
        fin = np.full(N, -1e20, np.float32)
        fin[-3:] = 0.0
        per_utt.append(P.arcs_to_graph_matrices(
            N, src, dst, il, rng.uniform(-1, 0, len(src)).astype(np.float32), fin))
    mats = [m.numpy() for m in P.batch_training_graph_matrices(per_utt)]
    assert mats[0].shape[2] <= 4 and mats[4].shape[2] <= 4
    lp = torch.log_softmax(torch.from_numpy(
        rng.standard_normal((T, B, C)).astype(np.float32)), -1).numpy()
    lens = np.array([45, 40, 31, 8], np.int32)
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    logZ, grad, zb = run_fwbw(lp, lens, mats, want_bwd=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    assert_posteriors(grad, want['grad'], lp, lens, mats, oracle_lib)
    np.testing.assert_allclose(zb, want['logZ_bwd'], rtol=RTOL_LOSS, atol=1e-4)


GROUPED = [
    ('mono_s49', 1, 49, {}, 60, 5),
    ('bi_s7', 2, 7, {}, 50, 6),
    ('bi_s7_ctxblank', 2, 7, dict(use_contextual_blanks=True), 40, 4),
    ('bi_s7_noself', 2, 7, dict(allow_nonblank_selfloops=False), 40, 4),
    ('bi_s49', 2, 49, {}, 14, 3),                   # the 2401-state x 51-arc CTC-G denominator
]


@pytest.mark.parametrize('name,order,S,kw,T,B', GROUPED, ids=[g[0] for g in GROUPED])
def test_grouped_decoding_graph_kernels(oracle_lib, name, order, S, kw, T, B):
    """Closed-form (group-factored) kernels for the decoding / denominator
    graphs against the oracle's generic sparse scan on the padded arc matrices
    of the SAME graph (fst_utils.py:662-676)."""
    from att_speech import _native, fst_utils as P
    rng = np.random.default_rng(len(name) * 7 + S)
    C = S ** order
    pg = P.CTCGraphGen(context_order=order, num_symbols=S, graph_build_args=kw)
    tagged = pg.get_decoding_matrices()
    assert tagged.grouped is not None and tagged[:4].grouped is not None
    mats = [m.numpy() for m in tagged]
    x = rng.standard_normal((T, B, C)).astype(np.float32) * 2
    lp = x - x.max(-1, keepdims=True)               # what get_fst_loss feeds (:479-484)
    lens = np.sort(rng.integers(max(1, T // 2), T + 1, size=B))[::-1].astype(np.int32).copy()
    lens[0] = T
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    d = dev()
    gg = _native.GroupedGraph(tagged.grouped, d)
    tl = torch.from_numpy(lens).to(d)
    logZ, grad, zb = _native.grouped_fwbw(torch.from_numpy(lp).to(d), tl, gg, -1e20,
                                          want_bwd_total=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logZ.cpu().numpy(), want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    assert_posteriors(grad.cpu().numpy(), want['grad'], lp, lens, mats, oracle_lib)
    np.testing.assert_allclose(zb.cpu().numpy(), want['logZ_bwd'], rtol=RTOL_LOSS, atol=1e-4)
    s, _ = _native.grouped_forward(torch.from_numpy(lp).to(d), tl, gg, -1e20)
    np.testing.assert_allclose(s.cpu().numpy(), want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    vs, vil = oracle_lib.path_forward(lp, lens, mats, viterbi=True)
    v, best = _native.grouped_forward(torch.from_numpy(lp).to(d), tl, gg, -1e20,
                                      viterbi=True, want_path=True)
    np.testing.assert_allclose(v.cpu().numpy(), vs, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(best.cpu().numpy(), vil)        # bit-exact labels
    # and through the reference-shaped surface: tagged matrices take the fast path
    lpt = torch.from_numpy(lp).to(d).requires_grad_()
    z = P.path_reduction(lpt, torch.from_numpy(lens), tagged, red_kind='logsumexp')
    z.sum().backward()
    assert_posteriors(lpt.grad.cpu().numpy(), want['grad'], lp, lens, mats, oracle_lib)


@pytest.mark.parametrize('order,S,kw', [(1, 49, {}), (2, 7, {}), (2, 49, {}),
                                        (2, 7, dict(use_contextual_blanks=True)),
                                        (2, 7, dict(allow_nonblank_selfloops=False))])
def test_device_graph_builder_equals_host_builder(order, S, kw):
    """asr_ctc_graph_build writes the same arcs as the host
    closed form (itself bit-equal to the mini-OpenFst composition,
    tests/test_graphs.py); K is fixed at 3 on the device."""
    from att_speech import fst_utils as P
    rng = np.random.default_rng(order * 31 + S)
    B, Lmax = 9, 12
    lens = np.array([12, 11, 9, 7, 5, 3, 2, 1, 0])
    labs = rng.integers(1, min(S, 5), size=(B, Lmax))
    labs[0, :5] = [2, 2, 2, 3, 3]
    pg = P.CTCGraphGen(context_order=order, num_symbols=S, graph_build_args=kw)
    host = pg.get_training_matrices_batch(labs, lens)
    g = pg.get_training_graph_device(labs, lens, dev())
    torch.cuda.synchronize()

    def pad3(m, fill):
        out = torch.full((m.shape[0], m.shape[1], 3), fill, dtype=m.dtype)
        out[:, :, :m.shape[2]] = m
        return out
    for got, want, fill in [(g.src_in, host[0], 0), (g.il_in, host[1], 0), (g.w_in, host[2], -1e20),
                            (g.dst_out, host[4], 0), (g.il_out, host[5], 0), (g.w_out, host[6], -1e20)]:
        np.testing.assert_array_equal(got.cpu().numpy(), pad3(want, fill).numpy().astype(got.cpu().numpy().dtype))
    np.testing.assert_array_equal(g.term.cpu().numpy(), host[3].squeeze(-1).numpy())
    # and it drives the scan: same loss as with the host-built tensors
    T = 40
    lp = torch.log_softmax(torch.from_numpy(
        rng.standard_normal((T, B, S ** order)).astype(np.float32)), -1).to(dev())
    tl = torch.full((B,), T, dtype=torch.int32)
    a = P.path_reduction(lp, tl, g)
    b2 = P.path_reduction(lp, tl, host)
    np.testing.assert_allclose(a.cpu().numpy(), b2.cpu().numpy(), rtol=1e-6)


def test_full_size_properties():
    """BASELINE shape (T'=334, C=49, L<=100) at a saturating batch: properties
    that need no oracle — per-frame posteriors sum to one, zero rows past the
    end, forward total == backward total, == torch's own CTC on the GPU."""
    from att_speech import fst_utils as P
    B, T, S = 256, 334, 49
    rng = np.random.default_rng(99)
    lens = np.array([T - (b % 64) for b in range(B)], np.int32)
    lens = np.sort(lens)[::-1].copy()
    llens = np.array([100 - 2 * (b % 16) for b in range(B)])
    labs = rng.integers(2, 49, size=(B, 100))
    lp = torch.log_softmax(torch.from_numpy(
        rng.standard_normal((T, B, S)).astype(np.float32)), -1)
    pg = P.CTCGraphGen(context_order=1, num_symbols=S)
    mats = [m.numpy() for m in pg.get_training_matrices_batch(labs, llens)]
    logZ, grad, zb = run_fwbw(lp.numpy(), lens, mats, want_bwd=True)
    np.testing.assert_allclose(zb, logZ, rtol=RTOL_LOSS)
    mask = (np.arange(T)[:, None] < lens[None, :])
    # posteriors of a frame sum to one up to the fp32 conditioning of a T'=334
    # step recurrence (the reference itself tolerates 1e-3, fst_utils.py:477)
    np.testing.assert_allclose(grad.sum(-1)[mask], 1.0, atol=1e-3)
    assert not grad[~mask].any()
    assert grad.min() >= 0.0
    want = torch.nn.functional.ctc_loss(
        lp.to(dev()), torch.from_numpy(labs).to(dev()), torch.from_numpy(lens).long(),
        torch.from_numpy(llens), reduction='none').cpu().numpy()
    np.testing.assert_allclose(-logZ, want, rtol=RTOL_LOSS)


def test_path_reduction_surface():
    """att_speech.fst_utils.path_reduction keeps the reference's dispatch,
    autograd behaviour and error checks (fst_utils.py:322-397, 482-485)."""
    from att_speech import fst_utils as P
    g = golden('lattice_mono.npz')
    d = dev()
    mats = to_t([g['gm%d' % i] for i in range(8)])
    lens = torch.from_numpy(g['lens'])
    lp = torch.from_numpy(g['lp']).to(d).requires_grad_()
    z = P.path_reduction(lp, lens, mats, red_kind='logsumexp')
    w = torch.from_numpy(g['w']).to(d)
    (z * w).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['fwbw_logZ'], rtol=RTOL_LOSS)
    # the gradient is the posterior scaled by the per-utterance weight w_b (|w| up to ~2 here)
    np.testing.assert_allclose(lp.grad.cpu().numpy(), g['fwbw_grad_w'],
                               atol=grad_atol(g['fwbw_logZ']) * max(1.0, float(np.abs(g['w']).max())))
    # 4 matrices + autodiff request -> still differentiable
    lp2 = torch.from_numpy(g['lp']).to(d).requires_grad_()
    z2 = P.path_reduction(lp2, lens, mats[:4], red_kind='logsumexp_autodiff')
    z2.sum().backward()
    np.testing.assert_allclose(lp2.grad.cpu().numpy(), g['autodiff_grad'], atol=1e-4)
    # viterbi: gradient is the one-hot best path; selidx recipe of the decoder
    lp3 = torch.from_numpy(g['lp']).to(d).requires_grad_()
    v = P.path_reduction(lp3, lens, mats[:4], red_kind='viterbi')
    (-v.sum()).backward()
    sel = lp3.grad.min(-1)[1].cpu().numpy()
    for b, l in enumerate(g['lens']):
        np.testing.assert_array_equal(sel[:l, b], g['viterbi_selidx'][:l, b])
    # unsorted lengths are refused like the reference (fst_utils.py:382,432)
    with pytest.raises(AssertionError):
        P.path_reduction(lp, torch.flip(lens, [0]), mats)
    # CPU tensors: no fallback
    from att_speech import _native
    with pytest.raises(_native.NativeLibraryError):
        P.path_reduction(torch.from_numpy(g['lp']), lens, mats)


def test_log_softmax_and_rowmax(oracle_lib):
    from att_speech import _native
    d = dev()
    g = golden('normalized_acts.npz')
    S = int(g['S'])
    x = torch.from_numpy(g['acts']).to(d)
    np.testing.assert_allclose(_native.log_softmax_fwd(x, S * S).cpu().numpy(),
                               g['zero'], atol=2e-6)
    np.testing.assert_allclose(_native.log_softmax_fwd(x, S).cpu().numpy(),
                               g['one'], atol=2e-6)
    rng = np.random.default_rng(3)
    for rows, group in [(1000, 49), (37, 2401), (5, 1), (64, 64), (3, 130), (2, 5000)]:
        a = (rng.standard_normal((rows, group)) * 4).astype(np.float32)
        ta = torch.from_numpy(a).to(d)
        y = _native.log_softmax_fwd(ta, group)
        # a few fp32 ulps at |y| ~ 20: the 2401-term sums are ordered differently
        np.testing.assert_allclose(y.cpu().numpy(), oracle_lib.log_softmax(a), rtol=1e-6, atol=1e-5)
        dy = torch.from_numpy(rng.standard_normal((rows, group)).astype(np.float32)).to(d)
        dx = _native.log_softmax_bwd(y, dy, group)
        tr = ta.clone().requires_grad_()
        torch.log_softmax(tr, -1).backward(dy)
        np.testing.assert_allclose(dx.cpu().numpy(), tr.grad.cpu().numpy(), rtol=1e-5, atol=2e-5)
    # advanced_decoder.py:479-484
    T, B, C = 50, 7, 49
    a = rng.standard_normal((T, B, C)).astype(np.float32)
    lens = np.array([50, 44, 30, 30, 9, 1, 0], np.int32)
    y, rmax, msum = _native.sub_rowmax(torch.from_numpy(a).to(d),
                                       torch.from_numpy(lens).to(d))
    np.testing.assert_array_equal(rmax.cpu().numpy(), a.max(-1))
    np.testing.assert_array_equal(y.cpu().numpy(), a - a.max(-1, keepdims=True))
    mask = np.arange(T)[:, None] < lens[None, :]
    np.testing.assert_allclose(msum.cpu().numpy(), (a.max(-1) * mask).sum(0), rtol=1e-5, atol=1e-5)


SORT_CASES = [
    # few symbols -> long runs of one label: runs > 64 lanes, runs crossing wave boundaries
    ('s3_long_runs', dict(order=1, S=3, T=230, B=5, Lmax=100, seed=11)),  # runs of ~100 lanes (the round-2 red case)
    ('s5_n301', dict(order=1, S=5, T=200, B=4, Lmax=150, seed=12)),            # H = 320
    ('s49_n255_no_slack', dict(order=1, S=49, T=140, B=6, Lmax=127, seed=13)),  # N = H - 1
    ('s49_n511', dict(order=1, S=49, T=300, B=3, Lmax=255, seed=14)),           # 8 waves per group
    ('s64_c_eq_lanes', dict(order=1, S=64, T=90, B=7, Lmax=31, seed=15)),       # C = H = 64
]


@pytest.mark.parametrize('name,kw', SORT_CASES, ids=[c[0] for c in SORT_CASES])
def test_label_sorted_segmented_sums(oracle_lib, name, kw):
    """The C <= H kernel sorts the states by label and forms the per-class posterior sums
    with a segmented scan over lanes (csrc/lattice.hip, FL == 1): label runs longer than a
    wave, runs that cross 16-lane rows and wave boundaries, no spare lanes, C == H."""
    lp, lens, mats = _random_case(**kw)
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    logZ, grad, zb = run_fwbw(lp, lens, mats, want_bwd=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    # long lattices with |logZ| in the hundreds: BOTH fp32 evaluations carry the conditioning
    # error of exp(alpha + beta - logZ) (the oracle's own rows sum to 1 +- 2e-4 here); the
    # fp64 evaluation arbitrates (assert_posteriors)
    assert_posteriors(grad, want['grad'], lp, lens, mats, oracle_lib)
    tol = grad_atol(want['logZ'])
    np.testing.assert_allclose(zb, want['logZ_bwd'], rtol=RTOL_LOSS, atol=1e-4)
    # every frame's posteriors sum to one over the classes
    T, B = lp.shape[:2]
    feasible = want['logZ'] > -1e19           # (too few frames for the labels + repeats otherwise)
    assert feasible.sum() >= 2
    mask = (np.arange(T)[:, None] < lens[None, :]) & feasible[None, :]
    np.testing.assert_allclose(grad.sum(-1)[mask], 1.0, atol=max(2e-4, tol))


def _band_graph(rng, B, N, C, weighted):
    """Random band lattices: state n entered from a random subset of {n, n-1, n-2} (always
    from n-1, so every state is reachable), one label per state, weights 0 or random <= 0,
    the last two states final."""
    from att_speech import fst_utils as P
    per_utt = []
    for b in range(B):
        nb = N if b == 0 else int(rng.integers(max(3, N // 2), N + 1))
        lab = rng.integers(0, C, size=nb)
        src, dst = [], []
        for n in range(nb):
            for d_ in (0, 1, 2):
                if n - d_ < 0:
                    continue
                if d_ == 1 or n == 0 or rng.random() < 0.6:
                    src.append(n - d_); dst.append(n)
        src, dst = np.array(src), np.array(dst)
        w = (-rng.random(len(src)) * 2).astype(np.float32) if weighted else np.zeros(len(src), np.float32)
        fin = np.full(nb, -1e20, np.float32)
        fin[-2:] = (-rng.random(2)).astype(np.float32) if weighted else 0.0
        per_utt.append(P.arcs_to_graph_matrices(nb, src, dst, lab[dst], w, fin))
    return [m.numpy() for m in P.batch_training_graph_matrices(per_utt)]


BAND_CASES = [
    ('ctc_even_T', dict(order=1, S=49, T=120, B=9, Lmax=40, seed=1), None),
    ('ctc_odd_T', dict(order=1, S=49, T=121, B=5, Lmax=33, seed=31), None),
    ('ctc_n255', dict(order=1, S=49, T=300, B=4, Lmax=127, seed=32), None),
    ('ctc_long_runs', dict(order=1, S=3, T=230, B=5, Lmax=100, seed=11), None),
    ('ctc_c64', dict(order=1, S=64, T=90, B=7, Lmax=31, seed=15), None),
    ('ctc_short_and_empty', dict(order=1, S=49, T=30, B=6, Lmax=8, seed=7, lens=[30, 30, 17, 4, 3, 0]), None),
    ('ctc_t_le_prefetch', dict(order=1, S=49, T=9, B=3, Lmax=3, seed=33), None),
    ('bigram_classes_s7', dict(order=2, S=7, T=70, B=6, Lmax=20, seed=34), None),
    ('random_band_unit', None, dict(B=5, N=77, C=23, T=91, weighted=False, seed=35)),
    ('random_band_weighted', None, dict(B=6, N=130, C=40, T=161, weighted=True, seed=36)),
    ('random_band_n256', None, dict(B=3, N=256, C=64, T=330, weighted=True, seed=37)),
]


@pytest.mark.parametrize('name,kw,gk', BAND_CASES, ids=[c[0] for c in BAND_CASES])
def test_band_kernel_linear_domain(oracle_lib, name, kw, gk):
    """asr_lattice_fwbw_band_f32 (csrc/lattice_band.inc): alpha / beta as rescaled fp32 numbers,
    one wave per direction, posterior rows normalised by their own totals.  Against the fp32
    oracle for logZ, the fp64 arbiter for the posteriors; rows that sum to one to a few ulp
    prove the linear-domain path (not its in-kernel log-domain fallback) produced them."""
    if kw is not None:
        lp, lens, mats = _random_case(**kw)
    else:
        rng = np.random.default_rng(gk['seed'])
        mats = _band_graph(rng, gk['B'], gk['N'], gk['C'], gk['weighted'])
        T, B = gk['T'], gk['B']
        lens = np.sort(rng.integers(T // 2, T + 1, size=B))[::-1].astype(np.int32).copy()
        lens[0] = T
        lp = torch.log_softmax(torch.from_numpy(
            rng.standard_normal((T, B, gk['C'])).astype(np.float32) * 2), -1).numpy()
    from att_speech import _native
    assert _native._host_band_check(to_t(mats))
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    logZ, grad, zb = run_fwbw(lp, lens, mats, want_bwd=True, band=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    np.testing.assert_allclose(zb, want['logZ_bwd'], rtol=RTOL_LOSS, atol=1e-4)
    assert_posteriors(grad, want['grad'], lp, lens, mats, oracle_lib)
    T, B = lp.shape[:2]
    lens = np.asarray(lens)
    for b in range(B):
        assert not grad[lens[b]:, b].any()
    # utterances the linear-domain path keeps: feasible, >= 4 frames, and with room to spare —
    # when the frames barely suffice for the states, nearly all of alpha's mass sits on states
    # that can no longer reach the end, the live part underflows relative to the frame maximum,
    # the kernel notices that its three values of Z disagree and redoes the utterance in the
    # log domain (seen: 183 states in 130 frames, 255 states in 127; asserted here only for
    # utterances with at least one frame per state)
    nstates = np.array([(np.asarray(mats[2][b if mats[2].shape[0] > 1 else 0]) > -1e19).any(-1).sum()
                        for b in range(B)])
    lin = (want['logZ'] > -1e19) & (lens >= 4) & (lens >= nstates)
    assert lin.sum() >= 1
    mask = (np.arange(T)[:, None] < lens[None, :]) & lin[None, :]
    np.testing.assert_allclose(grad.astype(np.float64).sum(-1)[mask], 1.0, atol=3e-6)
    # and the log-domain kernel (its own parity: the tests above) agrees with it: two fp32
    # evaluations, each within ~2x the fp32 oracle's own distance from fp64
    logZ2, grad2, _ = run_fwbw(lp, lens, mats, want_bwd=True, band=False)
    np.testing.assert_allclose(logZ, logZ2, rtol=RTOL_LOSS, atol=1e-5)
    f64 = oracle_lib.path_logsumexp_f64(lp, lens, mats)
    feas = f64['logZ'] > -1e19
    err32 = np.abs(want['grad'][:, feas] - f64['grad'][:, feas]).max()
    assert np.abs(grad2[:, feas] - grad[:, feas]).max() <= max(ATOL_GRAD, 4 * err32)


def test_band_kernel_falls_back_inside_the_launch(oracle_lib):
    """What the linear domain cannot hold is detected and redone by the log-domain body of
    the same launch: (a) emissions far above 0 (raw logits: overflow), (b) a transcript the
    model rules out with probabilities of e^-200 per frame (total underflow: Z == 0 at the
    meeting point), (c) a graph that is not a band (caller's tag wrong)."""
    kw = dict(order=1, S=49, T=64, B=4, Lmax=12, seed=41)
    lp, lens, mats = _random_case(**kw)
    # (a)
    big = (lp * -8.0 + 60.0).astype(np.float32)
    want = oracle_lib.path_logsumexp(big, lens, mats)
    logZ, grad, zb = run_fwbw(big, lens, mats, want_bwd=True, band=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS)
    assert_posteriors(grad, want['grad'], big, lens, mats, oracle_lib)
    # (b) every class but the blank at -200: all label emissions underflow in the linear domain
    peaky = np.full_like(lp, -200.0)
    peaky[..., 0] = 0.0
    want = oracle_lib.path_logsumexp(peaky, lens, mats)
    assert (want['logZ'] < -1000).sum() >= 2 and (want['logZ'] > -1e19).all()
    logZ, grad, zb = run_fwbw(peaky, lens, mats, want_bwd=True, band=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS)
    assert_posteriors(grad, want['grad'], peaky, lens, mats, oracle_lib)
    # (c) per-arc labels: not state-labelled
    from att_speech import fst_utils as P
    rng = np.random.default_rng(42)
    C, T, B, N = 11, 40, 3, 30
    per_utt = []
    for b in range(B):
        src, dst = [], []
        for n in range(N):
            for d_ in (0, 1, 2):
                if n - d_ >= 0:
                    src.append(n - d_); dst.append(n)
        src, dst = np.array(src), np.array(dst)
        fin = np.full(N, -1e20, np.float32)
        fin[-1] = 0.0
        per_utt.append(P.arcs_to_graph_matrices(
            N, src, dst, rng.integers(0, C, len(src)), np.zeros(len(src), np.float32), fin))
    mats = [m.numpy() for m in P.batch_training_graph_matrices(per_utt)]
    lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((T, B, C)).astype(np.float32)), -1).numpy()
    lens = np.array([40, 36, 30], np.int32)
    want = oracle_lib.path_logsumexp(lp, lens, mats)
    from att_speech import _native
    cnt = _native._band_counter(dev())[0]
    before = int(cnt.item())
    logZ, grad, zb = run_fwbw(lp, lens, mats, want_bwd=True, band=True)
    np.testing.assert_allclose(logZ, want['logZ'], rtol=RTOL_LOSS)
    assert_posteriors(grad, want['grad'], lp, lens, mats, oracle_lib)
    # ... and counted in the caller's running counter (all three utterances: wrong graph shape),
    # which the host policy reads without a sync and answers with a cool-down on the log-domain
    # kernel once more than a tenth of a batch was redone
    assert int(cnt.item()) - before == B
    st = _native._BAND_STATE
    torch.cuda.synchronize()
    assert st['pending'] is not None and not _native._band_policy_allows() and st['cool'] == _native._BAND_COOLDOWN - 1
    st.update(cool=0)


def test_full_size_bichar_numerator_properties():
    """The ctc_bi shape (C = 2401 bigram classes, T' = 334, L <= 100) at B = 64: per-frame
    posteriors sum to one, zeros past the end, forward total == backward total, and the
    bigram lattice with labels prev*49+cur collapses onto torch's mono CTC on repeat-free
    transcripts when every bigram class is given its last symbol's mono log-prob
    (lp_bi[c] = lp_mono[c % 49])."""
    from att_speech import fst_utils as P
    B, T, S = 64, 334, 49
    rng = np.random.default_rng(5)
    lens = np.sort(np.array([T - (b % 32) for b in range(B)], np.int32))[::-1].copy()
    llens = np.array([100 - 2 * (b % 16) for b in range(B)])
    mono = rng.integers(2, 49, size=(B, 100))
    for j in range(1, 100):      # repeat-free transcripts: a repeated symbol is where the bigram
        same = mono[:, j] == mono[:, j - 1]          # lattice and mono CTC differ
        mono[same, j] = 2 + (mono[same, j] - 2 + 1) % 47
    prev = np.concatenate([np.zeros((B, 1), mono.dtype), mono[:, :-1]], 1)
    labs = prev * S + mono
    lp_mono = torch.log_softmax(torch.from_numpy(rng.standard_normal((T, B, S)).astype(np.float32)), -1)
    lp = lp_mono.repeat(1, 1, S).contiguous()            # [T, B, 2401]: class c -> lp_mono[c % 49]
    pg = P.CTCGraphGen(context_order=2, num_symbols=S)
    mats = [m.numpy() for m in pg.get_training_matrices_batch(labs, llens)]
    logZ, grad, zb = run_fwbw(lp.numpy(), lens, mats, want_bwd=True)
    np.testing.assert_allclose(zb, logZ, rtol=RTOL_LOSS)
    mask = np.arange(T)[:, None] < lens[None, :]
    np.testing.assert_allclose(grad.sum(-1)[mask], 1.0, atol=1e-3)
    assert not grad[~mask].any() and grad.min() >= 0.0
    want = torch.nn.functional.ctc_loss(
        lp_mono.to(dev()), torch.from_numpy(mono).to(dev()), torch.from_numpy(lens).long(),
        torch.from_numpy(llens), reduction='none').cpu().numpy()
    np.testing.assert_allclose(-logZ, want, rtol=RTOL_LOSS)


def test_bichar_numerator_above_two_gib_stays_on_the_fast_kernel():
    """bi-char CTC at B = 768 per GPU: T' B C 4 = 2.46 GB > 2^31.  The state-labelled kernel
    addresses lp / grad with unsigned 32-bit byte offsets (limit 4 GiB, csrc/lattice.hip), so
    the batch stays on it; checked through the properties that do not need an oracle run of
    this size — posteriors of a frame sum to one, zeros past the end, forward total ==
    backward total — and against the SAME utterances run as small batches (whose offsets stay
    below 2^31): logZ and posteriors of the first, a middle and the last utterances, i.e.
    the ones whose bytes sit at both ends of the offset range."""
    from att_speech import _native, fst_utils as P
    d = dev()
    B, T, S = 768, 334, 49
    C = S * S
    assert T * B * C * 4 > 2 ** 31
    rng = np.random.default_rng(11)
    lens = np.sort(np.array([T - (b % 32) for b in range(B)], np.int32))[::-1].copy()
    llens = np.array([100 - 2 * (b % 16) for b in range(B)])
    mono = rng.integers(2, 49, size=(B, 100))
    prev = np.concatenate([np.zeros((B, 1), mono.dtype), mono[:, :-1]], 1)
    labs = prev * S + mono
    mats = P.CTCGraphGen(context_order=2, num_symbols=S).get_training_matrices_batch(labs, llens)
    gen = torch.Generator(device=d).manual_seed(3)
    lp = _native.log_softmax_fwd(torch.randn(T, B, C, device=d, generator=gen), C)
    lens_d = torch.from_numpy(lens).to(d)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')              # the slow-kernel warning must not fire
        logZ, grad, zb = _native.lattice_fwbw(lp, lens_d, _native.Graph(mats, d), -1e20, want_bwd_total=True)
    torch.cuda.synchronize()
    assert torch.isfinite(logZ).all() and (logZ > -1e19).all()
    assert (zb - logZ).abs().max().item() < 1e-3 * logZ.abs().max().item()
    rows = grad.sum(-1)                              # [T, B]
    mask = torch.arange(T, device=d)[:, None] < lens_d[None, :]
    # (log-domain fp32 at |alpha + beta| ~ 2700: one ulp of the exponent is 2.4e-4 of the posterior)
    assert (rows[mask] - 1.0).abs().max().item() < 5e-3
    assert not grad[~mask].any().item() and grad.min().item() >= 0.0
    for sel in ([0, 1, 2, 3], [382, 383, 384, 385], [764, 765, 766, 767]):
        idx = torch.tensor(sel, device=d)
        sub = [m[sel] if m.shape[0] == B else m for m in mats]
        z2, g2, _ = _native.lattice_fwbw(lp[:, idx].contiguous(), lens_d[idx], _native.Graph(sub, d), -1e20)
        torch.cuda.synchronize()
        np.testing.assert_allclose(logZ[idx].cpu().numpy(), z2.cpu().numpy(), rtol=1e-6)
        assert (grad[:, idx] - g2).abs().max().item() <= 2e-5
    del grad, lp
    torch.cuda.empty_cache()


def test_full_size_grouped_denominator_properties():
    """The CTC-G denominator (2401 states x 51 arcs, group-factored kernels) at T' = 334,
    B = 16: posteriors of a frame sum to one, zeros past the end, forward == backward total,
    the alpha-only scan gives the same total, and Viterbi <= logsumexp."""
    from att_speech import _native, fst_utils as P
    B, T, S = 16, 334, 49
    rng = np.random.default_rng(6)
    pg = P.CTCGraphGen(context_order=2, num_symbols=S)
    tagged = pg.get_decoding_matrices()
    d = dev()
    gg = _native.GroupedGraph(tagged.grouped, d)
    x = torch.from_numpy(rng.standard_normal((T, B, S * S)).astype(np.float32))
    lp = torch.log_softmax(x, -1).to(d)
    lens = np.sort(np.array([T - 7 * b for b in range(B)], np.int32))[::-1].copy()
    tl = torch.from_numpy(lens).to(d)
    logZ, grad, zb = _native.grouped_fwbw(lp, tl, gg, -1e20, want_bwd_total=True)
    s, _ = _native.grouped_forward(lp, tl, gg, -1e20)
    v, _ = _native.grouped_forward(lp, tl, gg, -1e20, viterbi=True, want_path=True)
    torch.cuda.synchronize()
    logZ, grad, zb, s, v = [a.cpu().numpy() for a in (logZ, grad, zb, s, v)]
    np.testing.assert_allclose(zb, logZ, rtol=RTOL_LOSS)
    np.testing.assert_allclose(s, logZ, rtol=RTOL_LOSS)
    assert (v <= logZ + 1e-3).all()
    mask = np.arange(T)[:, None] < lens[None, :]
    np.testing.assert_allclose(grad.sum(-1)[mask], 1.0, atol=1e-3)
    assert not grad[~mask].any() and grad.min() >= 0.0
    # locally normalised inputs: the sum over ALL label sequences of the decoding graph is <= 1
    assert (logZ <= 1e-3).all()


@pytest.mark.parametrize('T,B,C', [(23, 5, 49), (9, 3, 2401), (4, 2, 7), (50, 4, 130)])
def test_fused_normalise_and_shift(T, B, C):
    """asr_log_softmax_shift_{fwd,bwd}_f32 == log_softmax followed by the row-max subtraction
    (get_normalized_acts + advanced_decoder.py:479-484) and the gradient of that composition
    with the maximum detached"""
    from att_speech import _native
    rng = np.random.default_rng(T * 7 + C)
    x = torch.from_numpy(rng.standard_normal((T, B, C)).astype(np.float32) * 4)
    lens = torch.tensor(sorted(rng.integers(0, T + 1, size=B).tolist(), reverse=True), dtype=torch.int32)
    d = dev()
    y, nls, nls_sum = _native.log_softmax_shift_fwd(x.to(d), lens.to(d))
    xr = x.double().requires_grad_()
    lp = torch.log_softmax(xr, -1)
    mx = lp.max(-1, keepdim=True)[0].detach()
    want = lp - mx
    np.testing.assert_allclose(y.cpu().numpy(), want.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(nls.cpu().numpy(), mx.squeeze(-1).numpy(), rtol=1e-5, atol=1e-6)
    mask = (torch.arange(T)[:, None] < lens[None, :]).double()
    np.testing.assert_allclose(nls_sum.cpu().numpy(), (mx.squeeze(-1) * mask).sum(0).numpy(), rtol=1e-5, atol=1e-5)
    dy = torch.from_numpy(rng.standard_normal((T, B, C)).astype(np.float32))
    want.backward(dy.double())
    dx = _native.log_softmax_shift_bwd(y, nls, dy.to(d))
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=2e-5)


@pytest.mark.parametrize('order,S', [(1, 49), (2, 7)])
def test_negated_reduction_and_in_place_backward(order, S):
    """path_reduction(negate=True) = -path_reduction(): the kernels write the occupancies
    negated (asr_lattice_fwbw_signed_f32; the band kernel's small tensor is negated after the
    launch), PathLogSumExp.backward scales in place and asr_scale_rows_f32 leaves utterances
    with factor 1 alone.  Values and gradients against the plain form, with unit and non-unit
    incoming gradients."""
    from att_speech import fst_utils as P
    lp, lens, mats = _random_case(order=order, S=S, T=60, B=5, Lmax=14, seed=77)
    x = torch.from_numpy(lp).to(dev())
    tm = to_t(mats)
    w = torch.tensor([1.0, 1.0, -0.5, 1.0, 3.0], device=dev())
    for weights in (None, w):
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        ya = -P.path_reduction(a, torch.from_numpy(lens), tm)
        yb = P.path_reduction(b, torch.from_numpy(lens), tm, negate=True)
        np.testing.assert_array_equal(ya.detach().cpu().numpy(), yb.detach().cpu().numpy())
        (ya.sum() if weights is None else (ya * weights).sum()).backward()
        (yb.sum() if weights is None else (yb * weights).sum()).backward()
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        c = x.clone().requires_grad_(True)
        y = P.path_reduction(c, torch.from_numpy(lens), tm, negate=True).sum()
        y.backward(retain_graph=True)
        y.backward()


def test_band_kernel_from_labels_is_the_band_kernel_from_matrices(oracle_lib, monkeypatch):
    """Graphs built on the device from the transcripts (asr_ctc_graph_build, context order 1)
    carry the transcripts along: the band kernel then writes the CTC chain down from the labels
    instead of reading and checking the matrices.  Same bits as from the matrices, and the
    oracle's logZ on the host-built matrices of the same transcripts; repeated labels (no skip
    arc), an empty transcript and a one-label one included."""
    from att_speech import _native
    from att_speech import fst_utils as P
    rng = np.random.default_rng(91)
    B, T, S, Lmax = 7, 80, 49, 30
    llens = np.array([30, 21, 12, 30, 1, 0, 7], np.int64)
    labs = rng.integers(2, S, size=(B, Lmax))
    labs[3, 5:12] = labs[3, 5]                          # a run of one symbol
    lens = np.array([80, 80, 77, 75, 70, 64, 61], np.int32)
    lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((T, B, S)).astype(np.float32) * 2), -1)
    d = dev()
    g = _native.build_ctc_graph(torch.from_numpy(labs.astype(np.int32)).to(d),
                                torch.from_numpy(llens.astype(np.int32)).to(d), S, 1)
    assert g.ctc_labels is not None and g.band
    x, tl = lp.to(d), torch.from_numpy(lens).to(d)
    monkeypatch.setenv('ASR_LATTICE_BAND', '2')
    out = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('ASR_BAND_LABELS', flag)
        logZ, grad, zb = _native.lattice_fwbw(x, tl, g, -1e20, want_bwd_total=True)
        out[flag] = (logZ.cpu().numpy(), grad.cpu().numpy(), zb.cpu().numpy())
    for a, b_ in zip(out['1'], out['0']):
        np.testing.assert_array_equal(a, b_)
    mats = [m.numpy() for m in P.CTCGraphGen(context_order=1, num_symbols=S).get_training_matrices_batch(labs, llens)]
    want = oracle_lib.path_logsumexp(lp.numpy(), lens, mats)
    np.testing.assert_allclose(out['1'][0], want['logZ'], rtol=RTOL_LOSS, atol=1e-5)
    assert_posteriors(out['1'][1], want['grad'], lp.numpy(), lens, mats, oracle_lib)


def test_numerator_minus_denominator_node_equals_two_reductions():
    """fst_utils.NumeratorMinusDenominator (one gradient buffer: the numerator's negated
    occupancies, the denominator's added onto them by asr_lattice_grouped_fwbw_acc_f32) against
    -path_reduction(numerator) + path_reduction(decoding graph) as two autograd nodes: values
    bit-equal, gradients to fp32 rounding of one addition, ragged lengths (rows past an
    utterance's end stay zero) and non-unit incoming gradients."""
    from att_speech import fst_utils as P
    S, T, B, Lmax = 7, 50, 5, 9
    rng = np.random.default_rng(123)
    lens = np.array([50, 47, 40, 33, 21], np.int64)
    llens = np.array([9, 4, 7, 1, 3], np.int64)
    labs = rng.integers(1, S, size=(B, Lmax))
    gg = P.CTCGraphGen(context_order=2, num_symbols=S)
    num = gg.get_training_matrices_batch(labs, llens)
    den = gg.get_decoding_matrices('cpu')
    x = torch.from_numpy(rng.standard_normal((T, B, S * S)).astype(np.float32)).to(dev())
    x = x - x.max(-1, keepdim=True)[0]
    w = torch.tensor([1.0, 1.0, 2.0, 1.0, -0.5], device=dev())
    tl = torch.from_numpy(lens)
    grouped = P._device_grouped(den, x.device)
    assert grouped is not None
    for weights in (None, w):
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        la = P.path_reduction(a, tl, num, negate=True) + P.path_reduction(a, tl, den, neg_inf=gg.nc_weight)
        lb = P.NumeratorMinusDenominator.apply(b, tl, num, grouped, gg.nc_weight)[0]
        np.testing.assert_allclose(lb.detach().cpu().numpy(), la.detach().cpu().numpy(), rtol=1e-6, atol=1e-5)
        (la.sum() if weights is None else (la * weights).sum()).backward()
        (lb.sum() if weights is None else (lb * weights).sum()).backward()
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-6, atol=1e-7)
        for i in range(B):
            assert not b.grad[lens[i]:, i].any()
