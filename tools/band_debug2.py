"""Development aid: compare the band kernel's workspace rows with a numpy evaluation."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-asr_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import test_lattice_gpu as T
from att_speech import _native
kw = dict(order=1, S=49, T=int(sys.argv[1]), B=int(os.environ.get('DBG_B', '2')), Lmax=int(sys.argv[2]), seed=int(os.environ.get('DBG_SEED', '1')))
lp, lens, mats = T._random_case(**kw)
if len(sys.argv) > 3 and sys.argv[3] == 'n256':
    gk = dict(B=3, N=256, C=64, T=200, weighted=True, seed=37)
    rng = np.random.default_rng(gk['seed'])
    mats = T._band_graph(rng, gk['B'], gk['N'], gk['C'], gk['weighted'])
    Tn_, B_ = gk['T'], gk['B']
    lens = np.sort(rng.integers(Tn_ // 2, Tn_ + 1, size=B_))[::-1].astype(np.int32).copy()
    lens[0] = Tn_
    lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((Tn_, B_, gk['C'])).astype(np.float32) * 2), -1).numpy()
    print('lens', lens, 'states', [(np.asarray(mats[2][i]) > -1e19).any(-1).sum() for i in range(B_)])
d = torch.device('cuda:0')
g = _native.Graph(T.to_t(mats), d)
L = _native.lib()
Tn, B, C = lp.shape
lpt = torch.from_numpy(lp).to(d); lt = torch.from_numpy(np.asarray(lens, np.int32)).to(d)
logZ = torch.empty(B, device=d); grad = torch.empty_like(lpt); zb = torch.empty(B, device=d)
nbytes = L.asr_lattice_fwbw_workspace_bytes(Tn, B, C, g.N)
ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=d)
p = _native._p
_native.check(L.asr_lattice_fwbw_band_f32(p(lpt), Tn, B, C, p(lt), p(g.src_in), p(g.il_in), p(g.w_in), p(g.term),
    p(g.dst_out), p(g.il_out), p(g.w_out), g.N, g.Kin, g.Kout, g.Bg, -1e20, 1.0, p(logZ), p(grad), p(zb), p(ws), nbytes, None, None, None, 0,
    _native._stream()), 'band')
torch.cuda.synchronize()
ws = ws.cpu().numpy()
b = int(os.environ.get('DBG_UTT', '0')); Ln = int(lens[b]); m = Ln // 2
src, il, w, term = [np.asarray(x[b]) for x in mats[:4]]
N = src.shape[0]; lab = il[:, 0]
W = np.zeros((3, N))
for n in range(N):
    for k in range(src.shape[1]):
        if w[n, k] > -1e19: W[n - src[n, k], n] = np.exp(w[n, k])
tau = np.exp(np.maximum(term[:, 0], -1e20).astype(np.float64))
e = np.exp(lp[:Ln, b].astype(np.float64))
al = np.zeros((Ln + 1, N)); al[0, 0] = 1
for t in range(Ln):
    s = W[0] * al[t]; s[1:] += W[1, 1:] * al[t, :-1]; s[2:] += W[2, 2:] * al[t, :-2]
    al[t + 1] = s * e[t, lab]; al[t + 1] /= al[t + 1].max()
be = np.zeros((Ln + 1, N)); be[Ln] = tau
for t in range(Ln - 1, -1, -1):
    x = be[t + 1] * e[t, lab]
    s = W[0] * x; s[:-1] += W[1, 1:] * x[1:]; s[:-2] += W[2, 2:] * x[2:]
    be[t] = s / s.max()
wc0 = (g.N + 63) // 64 * 64
wreg = (Tn + 2) * (wc0 + 64)
N4 = (N + 3) // 4 * 4
reg = ws[b * wreg:(b + 1) * wreg]
kreg = reg[(Tn + 2) * wc0:].view(np.int32)
nbad = 0
for slot in range(Ln):
    row = reg[slot * N4: slot * N4 + N].astype(np.float64)
    kk = kreg[slot * 64: slot * 64 + (N + 3) // 4].astype(np.float64)
    exp = al[slot + 1] if slot < m else be[slot + 1]
    ok = (exp > 1e-200) & (row > 0)
    l2 = np.log2(row[ok]) - np.repeat(kk, 4)[:N][ok] - np.log2(exp[ok])
    dev = l2 - np.median(l2)
    badn = np.nonzero(ok)[0][np.abs(dev) > 1e-2]
    miss = np.nonzero((exp > 1e-30 * exp.max()) & (row <= 0))[0]
    if (len(badn) or len(miss)) and nbad < 25:
        nbad += 1
        print('slot %d (%s): off states %s dev %s | zero-but-expected %s | K lanes %s' % (
            slot, 'alpha' if slot < m else 'beta', badn.tolist()[:8], np.round(dev[np.abs(dev) > 1e-2][:8], 2).tolist(),
            miss.tolist()[:8], kk[:6].astype(int).tolist()))
print('done', logZ.cpu().numpy())
for bb in range(B):
    print('utt', bb, 'fallback reason', int(ws[(bb + 1) * wreg - 1].view(np.int32)))
for slot in [int(x) for x in os.environ.get('DBG_SLOTS', '').split(',') if x]:
    row = reg[slot * N4: slot * N4 + 12].astype(np.float64)
    kk = kreg[slot * 64: slot * 64 + 3]
    exp = (al[slot + 1] if slot < m else be[slot + 1])[:12]
    print('slot', slot, 'K', kk.tolist())
    print('   log2 v   ', np.round(np.log2(np.maximum(row, 1e-300)), 2).tolist())
    print('   log2 true', np.round(np.log2(np.maximum(row, 1e-300)) - np.repeat(kk, 4), 2).tolist())
    print('   log2 want', np.round(np.log2(np.maximum(exp, 1e-300)), 2).tolist())
for d_ in (0, 1):
    rec = reg[Tn * wc0 + 64 + d_ * 16: Tn * wc0 + 64 + d_ * 16 + 16]
    print('dir', d_, 'par,lane,kx,ky,F0', rec[:5].tolist(), 'log2 x', np.round(np.log2(np.maximum(rec[5:9], 1e-300)), 1).tolist(),
          'log2 y', np.round(np.log2(np.maximum(rec[9:13], 1e-300)), 1).tolist(), 'q', rec[13:16].tolist())
