"""asr_lstm_wgrad_bf16 alone (profiling target):  python tools/bench_wgrad.py [B] [T]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 334
H = 320
dev = torch.device('cuda:0')
dg = torch.randn(T, B, 2, 4 * H, device=dev).to(torch.bfloat16)
x = torch.randn(T * B, H, device=dev).to(torch.bfloat16)
y = torch.randn(2, T + 2, B, H, device=dev).to(torch.bfloat16)
for _ in range(3):
    _native.lstm_wgrad(dg, x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    _native.lstm_wgrad(dg, x, y)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print('B=%d T=%d: %.1f us, %.1f TFLOP/s' % (B, T, us, 2.0 * T * B * 8 * H * 2 * H / us * 1e-6))
