"""Development aid: per-phase s_memtime cycles of the band lattice kernel (a -DBAND_STAMPS build of
csrc/lattice.hip, loaded through ASR_AMD_LIB).  Columns: prologue | phase 0 | meeting | phase 1 | tail,
per wave (A chain, B chain, A helper, B helper), averaged over the utterances."""
import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'pytorch-asr_amd'), ROOT, os.path.join(ROOT, 'tools')]
from att_speech import _native
import bench_lattice as bl
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T = 334
lens, mats, C, ns, na = bl.make(1, B, T, 'num')
d = torch.device('cuda:0')
g = _native.Graph(mats, d)
lp = _native.log_softmax_fwd(torch.randn(T, B, C, device=d), C)
tl = torch.from_numpy(lens).to(d)
L = _native.lib()
logZ = torch.empty(B, device=d); grad = torch.empty_like(lp)
nbytes = L.asr_lattice_fwbw_workspace_bytes(T, B, C, g.N)
ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=d)
p = _native._p
for _ in range(3):
    _native.check(L.asr_lattice_fwbw_band_f32(p(lp), T, B, C, p(tl), p(g.src_in), p(g.il_in), p(g.w_in), p(g.term),
        p(g.dst_out), p(g.il_out), p(g.w_out), g.N, g.Kin, g.Kout, g.Bg, -1e20, 1.0, p(logZ), p(grad), None, p(ws), nbytes, None, None, None, 0,
        _native._stream()), 'band')
torch.cuda.synchronize()
wc0 = (g.N + 63) // 64 * 64
wreg = (T + 2) * (wc0 + 64)
w = ws.cpu().numpy()[:B * wreg].reshape(B, wreg)
raw = w[:, T * wc0:T * wc0 + 32].reshape(B, 4, 8)
st = raw[:, :, :5]
hw = raw[:, :, 5].copy().view(np.int32); xcc = raw[:, :, 6].copy().view(np.int32) & 15
why = w[:, wreg - 1].copy().view(np.int32)
print('B=%d  fallbacks: %d' % (B, int((why != 0).sum())))
for i, name in enumerate(['A chain ', 'B chain ', 'A helper', 'B helper']):
    m = st[:, i].mean(0)
    print('%s prologue %7.0f | phase0 %7.0f (%4.0f/step) | meet %6.0f | phase1 %7.0f (%4.0f/step) | tail %6.0f' % (
        name, m[0], m[1], m[1] / (T // 2), m[2], m[3], m[3] / (T - T // 2), m[4]))

simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
from collections import Counter
load = Counter()
for b_ in range(B):
    for role in (0, 1):
        load[(int(cuid[b_, role]), int(simd[b_, role]))] += 1
print('chain waves per (CU, SIMD): histogram', sorted(Counter(load.values()).items()))
print('workgroups per CU histogram', sorted(Counter(Counter(cuid[:, 0].tolist()).values()).items()))
blk = {}
for b_ in range(min(B, 2048)):
    blk.setdefault(int(cuid[b_, 0]), []).append(b_)
print('utterances sharing a CU (first 4 CUs):', list(blk.items())[:4])
print('simd of roles 0..3, first 6 utterances:', simd[:6].tolist())
