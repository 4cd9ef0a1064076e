import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tools'))
from att_speech import _native
import bench_lattice as bl
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lens, mats, C, ns, na = bl.make(1, B, 334, 'num')
dev = torch.device('cuda:0')
g = _native.Graph(mats, dev)
lp = _native.log_softmax_fwd(torch.randn(334, B, C, device=dev), C)
tl = torch.from_numpy(lens).to(dev)
for _ in range(3):
    z, grad, _ = _native.lattice_fwbw(lp, tl, g)
torch.cuda.synchronize()
st = grad[167 if os.environ.get('CHAIN') else -1, :, :4].cpu().numpy()
print('B=%d cycles: phase0 %.0f  mid %.0f  phase1 %.0f  tail %.0f  (per step: %.0f / %.0f)' % (
    (B,) + tuple(st.mean(0)) + (st[:, 0].mean() / 167, st[:, 2].mean() / 167)))
