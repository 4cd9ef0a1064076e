"""Forward recurrence of one encoder layer (F == H): library GEMM + asr_lstm_bidir_fwd_bf16
against asr_lstm_bidir_fwd_fused_bf16; backward recurrence for reference.
  python tools/bench_lstm.py [B] [T] [H]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 334
H = int(sys.argv[3]) if len(sys.argv) > 3 else 320
dev = torch.device('cuda:0')
x = torch.randn(T, B, H, device=dev).to(torch.bfloat16)
wih = (torch.randn(8 * H, H, device=dev) * 0.05).to(torch.bfloat16)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.05).to(torch.bfloat16)
whhT = whh.transpose(1, 2).contiguous()
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
dy = torch.randn(T, B, H, device=dev)


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


def gemm_path():
    gx = torch.mm(x.view(T * B, H), wih.t()).view(T, B, 2, 4 * H)
    return _native.lstm_bidir_fwd(gx, whh, lens, want_y=False)


def fused_path():
    return _native.lstm_bidir_fwd_fused(x, wih, whh, lens, want_y=False)


gx = torch.mm(x.view(T * B, H), wih.t()).view(T, B, 2, 4 * H)
out = _native.lstm_bidir_fwd(gx, whh, lens)
print('B=%d T=%d H=%d' % (B, T, H))
print('gemm only              %8.1f us' % timed(lambda: torch.mm(x.view(T * B, H), wih.t())))
print('recurrence (gx given)  %8.1f us' % timed(lambda: _native.lstm_bidir_fwd(gx, whh, lens, want_y=False)))
print('gemm + recurrence      %8.1f us' % timed(gemm_path))
if _native.lstm_fused_supported(B, H):
    print('fused                  %8.1f us' % timed(fused_path))
print('backward recurrence    %8.1f us' % timed(lambda: _native.lstm_bidir_bwd(dy, whhT, lens, out[2], out[3])))
dgb = _native.lstm_bidir_bwd(dy, whhT, lens, out[2], out[3])
wk = wih.t().contiguous().t()
print('dx gemm only           %8.1f us' % timed(lambda: torch.mm(dgb.view(T * B, 8 * H), wk, out_dtype=torch.float32)))
if _native.lstm_fused_supported(B, H, backward=True):
    wihT = wih.view(2, 4 * H, H).transpose(1, 2).contiguous()
    print('backward fused         %8.1f us' % timed(lambda: _native.lstm_bidir_bwd_fused(dy, whhT, wihT, lens, out[2], out[3])))
    dxp = _native.lstm_bidir_bwd_fused(dy, whhT, wihT, lens, out[2], out[3])[1]
    print('backward, dy as planes %8.1f us' % timed(lambda: _native.lstm_bidir_bwd(dxp, whhT, lens, out[2], out[3], planes=True)))
_native.lstm_check_errors()
if _native.lstm_wgrad_supported(H):
    from att_speech.modules.encoders import native_lstm
    xb2 = x.view(T * B, H)
    print('weight grads, library  %8.1f us' % timed(lambda: native_lstm._weight_gradients_library(dgb, xb2, out[1], T, B, H, H)))
    print('weight grads, kernel   %8.1f us' % timed(lambda: _native.lstm_wgrad(dgb, xb2, out[1])))
    print('  dW_hh only           %8.1f us' % timed(lambda: _native.lstm_wgrad(dgb, None, out[1])))
if _native.lstm_dgrad_supported(H):
    print('dx kernel              %8.1f us' % timed(lambda: _native.lstm_dgrad(dgb, wih)))
