"""Development aid: first-convolution kernels at the WSJ width (Fo = 38), one and three channels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'pytorch-asr_amd'), ROOT]
from att_speech import _native
d = torch.device('cuda:0')
B, T, F = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 1000, 81


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin in (1, 3):
    x = torch.randn(B, T, F, cin, device=d) if cin > 1 else torch.randn(B, T, F, device=d)
    w = torch.randn(32, cin, 7, 7, device=d) * 0.1
    y = _native.conv1_fwd(x, w)
    out_mb = y.numel() * 2 / 1e6
    for sums in (False, True):
        us = timed(lambda: _native.conv1_fwd(x, w, want_sums=sums))
        print('cin=%d sums=%d fwd %7.1f us  (out %.0f MB, in %.0f MB -> %.2f TB/s)' % (
            cin, sums, us, out_mb, x.numel() * 4 / 1e6, (out_mb + x.numel() * 4 / 1e6) / us))
    dy = torch.randn_like(y)
    us = timed(lambda: _native.conv1_wgrad(x, dy))
    print('cin=%d wgrad %7.1f us' % (cin, us))
