import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, H = 64, 320
dev = torch.device('cuda:0')
gx = torch.randn(T, B, 2, 4 * H, device=dev)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.05).to(torch.bfloat16)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(3):
    y, ybf, gates, cs = _native.lstm_bidir_fwd(gx, whh, lens)
torch.cuda.synchronize()
bt = 64 if B >= 256 else 32
v = y[T // 2].reshape(B, 2, H)[0::bt]          # rows b0 of every workgroup
v = v.reshape(-1, 2, H // 32, 32)[..., :4]     # [btile, dir, jtile, 4 stamps]
print('B=%d cycles: frag-issue %.0f  gemm-done %.0f  lds+barrier %.0f  pointwise %.0f' % (
    (B,) + tuple(v.reshape(-1, 4).mean(0).tolist())))
