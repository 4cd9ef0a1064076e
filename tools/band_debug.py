"""Development aid: where do the band kernel's posteriors differ from the fp64 oracle?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-asr_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import test_lattice_gpu as T
from oracle import oracle
kw = dict(order=1, S=49, T=int(sys.argv[1]), B=int(os.environ.get('DBG_B', '2')), Lmax=int(sys.argv[2]), seed=int(os.environ.get('DBG_SEED', '1')))
lp, lens, mats = T._random_case(**kw)
print('lens', lens, 'N', mats[0].shape)
want = oracle.path_logsumexp_f64(lp, lens, mats)
logZ, grad, zb = T.run_fwbw(lp, lens, mats, want_bwd=True, band=True)
print('logZ', logZ, want['logZ'])
print('zb  ', zb, want['logZ_bwd'])
nanpos = np.argwhere(~np.isfinite(grad))
print('non-finite entries', len(nanpos), nanpos[:12].tolist())
if len(nanpos):
    import collections
    print('by utterance', collections.Counter(nanpos[:, 1].tolist()), 'frames', sorted(set(nanpos[:, 0].tolist()))[:20], 'classes', sorted(set(nanpos[:, 2].tolist()))[:20])
print('rowsum-1 per utterance', [float(np.abs(np.nan_to_num(grad[:lens[b_], b_]).sum(-1) - 1).max()) for b_ in range(lp.shape[1])])
err = np.abs(np.nan_to_num(grad, nan=9.0) - want['grad'])
for b in range(lp.shape[1]):
    print('utt', b, 'len', lens[b])
    for t in range(lp.shape[0]):
        e = err[t, b]
        if e.max() > 1e-4 and t % int(sys.argv[3]) == 0:
            c = int(e.argmax())
            print('  t=%d maxerr %.4f at class %d  got %.5f want %.5f  rowsum got %.5f' % (
                t, e.max(), c, grad[t, b, c], want['grad'][t, b, c], grad[t, b].sum()))
b = 0
lab = np.asarray(mats[1][b][:, 0])
print('labels of states', lab.tolist())
for t in [int(x) for x in sys.argv[4:]]:
    print('frame', t)
    for c in range(lp.shape[2]):
        if grad[t, b, c] > 1e-6 or want['grad'][t, b, c] > 1e-6:
            print('   class %2d got %.6f want %.6f ratio %.4f  states %s' % (
                c, grad[t, b, c], want['grad'][t, b, c], grad[t, b, c] / max(want['grad'][t, b, c], 1e-30),
                np.nonzero(lab == c)[0].tolist() if c else 'blank'))
