"""CPU probe for the next lattice-kernel design: the alpha/beta recurrences
in the LINEAR domain with power-of-two rescaling instead of log-sum-exp per arc.

Per state and frame the log-domain scan costs 3 v_exp + 1 v_log (quarter-rate, 64 of the
~160 VALU cycles of a step, and the kernel is VALU-issue-bound at two workgroups per CU);
the scaled linear recurrence  a'[n] = e[n] * sum_k w_k a[src_k]  needs none (one exp per
emission, or 49 per frame if done per class).  This script restates that arithmetic in
numpy float32 in two variants and compares logZ and the gradient with the float64
log-domain result (itself checked against the golden vectors):
  * shared scale: emissions shifted by the row maximum, the state vector rescaled by the
    exact power of two of its maximum every R frames.  Fine for flat posteriors, WRONG for
    peaked ones: states more than 2^-126 below the frame's maximum are flushed although
    they can carry the best path later (and a whole vector can underflow between rescales);
  * mantissa + exponent per value (block floating point with an int32 exponent): the
    dynamic range of the log domain, no transcendental per arc — per state and frame one
    exponent max3, three v_ldexp, three FMAs, one v_frexp pair; the emission is
    2^(lp*log2e) split into floor and exp2(fraction).  Matches float64 to 1e-9 / 1e-6 at
    every peaking tested: the arithmetic to build the next scan kernel on.

  python tools/linear_domain_probe.py            # golden lattices + a bench-shape case
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEG = -1e20


def log_domain_f64(lp, T, src, il, w, term):
    """reference: alpha/beta in float64 log domain, gradient w.r.t. lp [T,C]"""
    N, K = src.shape
    a = np.full((T + 1, N), -np.inf); a[0, 0] = 0.0
    ww = np.where(w > NEG / 2, w.astype(np.float64), -np.inf)
    for t in range(T):
        x = a[t][src] + ww + lp[t][il]
        a[t + 1] = np.logaddexp.reduce(x, axis=1)
    tt = np.where(term > NEG / 2, term.astype(np.float64), -np.inf)
    logZ = np.logaddexp.reduce(a[T] + tt)
    b = np.full((T + 1, N), -np.inf); b[T] = tt
    grad = np.zeros_like(lp, dtype=np.float64)
    for t in range(T - 1, -1, -1):
        x = a[t][src] + ww + lp[t][il] + b[t + 1][:, None]          # arc posteriors (in-arc form)
        post = np.exp(x - logZ)
        np.add.at(grad[t], il.ravel(), post.ravel())
        nb = np.full(N, -np.inf)
        np.logaddexp.at(nb, src.ravel(), (ww + lp[t][il] + b[t + 1][:, None]).ravel())
        b[t] = nb
    return logZ, grad


def linear_domain_f32(lp, T, src, il, w, term, R=8):
    """what the kernel would do, float32; returns (logZ, grad, underflow flag)"""
    f = np.float32
    N, K = src.shape
    C = lp.shape[1]
    lw = np.where(w > NEG / 2, np.exp(w.astype(np.float64)), 0.0).astype(f)       # arc weights
    lt = np.where(term > NEG / 2, np.exp(term.astype(np.float64)), 0.0).astype(f)
    rmax = lp[:T].max(1).astype(f)
    e = np.exp((lp[:T] - rmax[:, None]).astype(f)).astype(f)                         # [T,C], <= 1
    a = np.zeros((T + 1, N), f); a[0, 0] = 1
    ascale = np.zeros(T + 1, np.int64)                       # log2 scale accumulated up to frame t
    for t in range(T):
        v = (a[t][src] * lw * e[t][il]).astype(f).sum(1, dtype=f)
        s = 0
        if (t + 1) % R == 0:
            m = v.max()
            if m > 0:
                s = int(np.frexp(m)[1])
                v = np.ldexp(v, -s).astype(f)                # exact power-of-two rescale
        a[t + 1] = v
        ascale[t + 1] = ascale[t] + s
    zt = (a[T] * lt).sum(dtype=f)
    under = not (zt > 0 and np.isfinite(zt))
    log2e = 1.4426950408889634
    logZ = (np.log2(np.float64(zt)) + ascale[T]) / log2e + rmax.astype(np.float64).sum() if not under else -np.inf
    b = np.zeros((T + 1, N), f); b[T] = lt
    bscale = np.zeros(T + 1, np.int64)
    grad = np.zeros((T, C), f)
    for t in range(T - 1, -1, -1):
        arc = (lw * e[t][il] * b[t + 1][:, None]).astype(f)                        # [N,K]
        # posterior of arc k into n: a_t[src] * arc / Z with the scales of a_t, b_{t+1} and Z
        sh = ascale[t] + bscale[t + 1] - ascale[T]
        post = (a[t][src] * arc).astype(f) * f(1.0 / zt) if not under else np.zeros_like(arc)
        post = np.ldexp(post, int(sh)).astype(f)
        np.add.at(grad[t], il.ravel(), post.ravel())
        nb = np.zeros(N, f)
        np.add.at(nb, src.ravel(), arc.ravel())
        s = 0
        if t % R == 0:
            m = nb.max()
            if m > 0:
                s = int(np.frexp(m)[1])
                nb = np.ldexp(nb, -s).astype(f)
        b[t] = nb
        bscale[t] = bscale[t + 1] + s
    return logZ, grad, under


def bfp_domain_f32(lp, T, src, il, w, term):
    """Same recurrences with every value kept as (float32 mantissa in [0.5,1), int32
    exponent): the dynamic range of the log domain (no shared scale, nothing to underflow)
    without a transcendental per arc.  Per state and frame: exponent max3, three v_ldexp,
    three FMAs, one v_frexp pair; the emission enters as 2^(lp*log2e) split into an integer
    exponent and exp2 of the fraction (ONE v_exp per class and frame)."""
    f = np.float32
    N, K = src.shape
    C = lp.shape[1]
    EMIN = np.int64(-(1 << 40))                               # exponent of an exact zero

    def split(x64):                                          # log-domain float64 -> (mant f32, exp)
        fin = np.isfinite(x64)
        l2 = np.where(fin, x64, 0.0) * 1.4426950408889634
        ex = np.floor(l2)
        m = np.exp2((l2 - ex)).astype(f)                     # [1,2): v_exp_f32 of the fraction
        return np.where(fin, m, f(0)), np.where(fin, ex, EMIN).astype(np.int64)

    wm, we = split(np.where(w > NEG / 2, w.astype(np.float64), -np.inf))
    tm, te = split(np.where(term > NEG / 2, term.astype(np.float64), -np.inf))
    em, ee = split(lp[:T].astype(np.float64))                 # [T,C]

    def norm(m, e):                                          # v_frexp_mant / v_frexp_exp
        mm, k = np.frexp(m)
        return mm.astype(f), np.where(m > 0, e + k, EMIN)

    def combine(ms, es):                                     # sum_k ms[...,k] * 2^es[...,k]
        emax = es.max(-1, keepdims=True)
        sh = np.maximum(es - emax, -200)                      # v_ldexp clamps to zero far below
        acc = np.ldexp(ms, sh.astype(np.int64)).astype(f).sum(-1, dtype=f)
        return norm(acc, emax[..., 0])

    am = np.zeros((T + 1, N), f); ae = np.full((T + 1, N), EMIN); am[0, 0] = 0.5; ae[0, 0] = 1
    for t in range(T):
        ms = (am[t][src] * wm * em[t][il]).astype(f)
        es = ae[t][src] + we + ee[t][il]
        es = np.where(ms > 0, es, EMIN)
        am[t + 1], ae[t + 1] = combine(ms, es)
    zm, ze = combine((am[T] * tm).astype(f)[None, :], np.where(am[T] * tm > 0, ae[T] + te, EMIN)[None, :])
    zm, ze = zm[0], ze[0]
    logZ = (np.log2(np.float64(zm)) + ze) / 1.4426950408889634
    bm = np.zeros((T + 1, N), f); be = np.full((T + 1, N), EMIN)
    bm[T], be[T] = tm, te
    grad = np.zeros((T, C), f)
    rz = f(1.0) / zm
    for t in range(T - 1, -1, -1):
        arcm = (wm * em[t][il] * bm[t + 1][:, None]).astype(f)
        arce = we + ee[t][il] + be[t + 1][:, None]
        arce = np.where(arcm > 0, arce, EMIN)
        pm = (am[t][src] * arcm).astype(f) * rz
        pe = np.maximum(ae[t][src] + arce - ze, -200)
        post = np.ldexp(pm, np.where(pm > 0, pe, 0).astype(np.int64)).astype(f)      # one v_ldexp
        np.add.at(grad[t], il.ravel(), post.ravel())
        # beta_t[s] = sum over the arcs leaving s: scatter form here, out-arc gather in a kernel
        nbm = np.zeros(N, f); nbe = np.full(N, EMIN)
        order = np.argsort(src.ravel(), kind='stable')
        ss, mm_, ee_ = src.ravel()[order], arcm.ravel()[order], arce.ravel()[order]
        for s_ in np.unique(ss):
            sel = ss == s_
            m1, e1 = combine(mm_[sel][None, :], ee_[sel][None, :])
            nbm[s_], nbe[s_] = m1[0], e1[0]
        bm[t], be[t] = nbm, nbe
    return logZ, grad, False


def compare(name, lp, lens, mats, want=None):
    res = {}
    for bi in range(lp.shape[1]):
        T = int(lens[bi])
        src, il, w, term = mats[0][bi], mats[1][bi], mats[2][bi], mats[3][bi][:, 0]
        z64, g64 = log_domain_f64(lp[:, bi].astype(np.float64), T, src, il, w, term)
        if want is not None:
            assert abs(z64 - want['logZ'][bi]) <= 2e-5 * max(1.0, abs(z64)), (name, bi)
        for tag, fn in (('shared scale, R=8', linear_domain_f32), ('mantissa+exponent', bfp_domain_f32)):
            z, g, under = fn(lp[:, bi], T, src, il, w, term)
            r = res.setdefault(tag, [0.0, 0.0, 0])
            if under or not np.isfinite(z):
                r[2] += 1
                continue
            r[0] = max(r[0], abs(z - z64) / max(1.0, abs(z64)))
            r[1] = max(r[1], float(np.abs(g - g64[:T]).max()))
    for tag, (dl, dg, nu) in res.items():
        print('%-26s %-18s max rel |dlogZ| %.1e  max |dgrad| %.1e  underflowed %d/%d'
              % (name, tag, dl, dg, nu, lp.shape[1]))


def main():
    gd = os.path.join(ROOT, 'tests', 'golden')
    for name in ('lattice_mono', 'lattice_bigram_s7'):
        p = os.path.join(gd, name + '.npz')
        if not os.path.exists(p):
            continue
        g = np.load(p)
        compare(name, g['lp'], g['lens'], [g['gm%d' % i] for i in range(4)],
                {'logZ': g['fwbw_logZ']})
    # bench shape: T'=334, L=100 (N=201), C=49, peaked and flat log-probs
    sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
    sys.path.insert(0, ROOT)
    from att_speech import fst_utils                     # host graph builder (no GPU needed)
    gg = fst_utils.CTCGraphGen(context_order=1, num_symbols=49)
    rng = np.random.default_rng(3)
    for scale in (1.0, 6.0, 20.0, 60.0):         # flat ... sharply peaked (wrong) posteriors
        T, B, C, L = 167, 2, 49, 50
        x = rng.standard_normal((T, B, C)).astype(np.float32) * scale
        mx = x.max(-1, keepdims=True)
        lp = (x - mx - np.log(np.exp(x - mx).sum(-1, keepdims=True))).astype(np.float32)
        labs = rng.integers(2, 49, size=(B, L))
        mats = [np.asarray(m) for m in gg.get_training_matrices_batch(labs, np.full(B, L))]
        compare('bench shape, logits x%.0f' % scale, lp, np.full(B, T), mats)


if __name__ == '__main__':
    main()
