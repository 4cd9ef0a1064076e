"""Convolution micro-benchmark (development tool): hand-written 7x7 32->32 kernels vs torch /
MIOpen on the bench shape [B, 32, 1006, 17] bf16 channels-last, stride (3, 1)."""
import os
import sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'pytorch-asr_amd', 'miopen_db'))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native          # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
    dev = torch.device('cuda:0')
    x = torch.randn(B, 32, 1006, 17, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(32, 32, 7, 7, device=dev) * 0.05
    wb = w.to(torch.bfloat16)
    flops = 2.0 * B * 334 * 11 * 32 * 32 * 49
    t = timeit(lambda: _native.conv7x7c32_fwd(x, w, 3))
    print('native fwd   %8.1f us  %6.1f TFLOP/s' % (t, flops / t / 1e6))
    t = timeit(lambda: F.conv2d(x, wb, None, (3, 1)))
    print('torch  fwd   %8.1f us  %6.1f TFLOP/s' % (t, flops / t / 1e6))
    if hasattr(_native, 'conv7x7c32_bwd_data'):
        y = _native.conv7x7c32_fwd(x, w, 3)
        dy = torch.randn_like(y)
        t = timeit(lambda: _native.conv7x7c32_bwd_data(dy, w, 1006, 17, 3))
        print('native dgrad %8.1f us  %6.1f TFLOP/s' % (t, flops / t / 1e6))
        t = timeit(lambda: torch.ops.aten.convolution_backward(
            dy, x, wb, None, (3, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
        print('torch  dgrad %8.1f us' % t)
    if hasattr(_native, 'conv7x7c32_wgrad'):
        y = _native.conv7x7c32_fwd(x, w, 3)
        dy = torch.randn_like(y)
        t = timeit(lambda: _native.conv7x7c32_wgrad(x, dy, 3))
        print('native wgrad %8.1f us  %6.1f TFLOP/s' % (t, flops / t / 1e6))
        t = timeit(lambda: torch.ops.aten.convolution_backward(
            dy, x, wb, None, (3, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False)))
        print('torch  wgrad %8.1f us' % t)


if __name__ == '__main__':
    main()
