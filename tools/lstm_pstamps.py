"""Per-phase cycle stamps of the persistent LSTM kernels (dev probe).
Build first (here, the .so travels to the GPU box):
  cd pytorch-asr_amd/csrc && for f in lattice lattice_grouped graph_build softmax lstm; do \
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DASR_LSTM_STAMPS -c $f.hip -o /tmp/st_$f.o; done; \
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c capi.cpp -o /tmp/st_capi.o; \
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_scratch/libasr_amd_stamps.so /tmp/st_*.o
Run:  ASR_AMD_LIB=gpurun_scratch/libasr_amd_stamps.so python tools/lstm_pstamps.py [B] [T]
Phases (thread 0 of every workgroup, averaged over steps and workgroups):
  0 gx/gates prefetch issue + team wait   1 barrier + hand-off tile load -> LDS + barrier
  2 MFMA + tile -> LDS + barrier          3 cell math + stage -> LDS + barrier + sc1 store issue
  4 store drain (vmcnt 0) + barrier       5 counter add + bulk stores issue
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 334
H = 320
dev = torch.device('cuda:0')
gx = torch.randn(T, B, 2, 4 * H, device=dev)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.05).to(torch.bfloat16)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
dy = torch.randn(T, B, 2, H, device=dev)
whhT = whh.view(2, 4 * H, H).transpose(1, 2).contiguous()
FUSED = os.environ.get('FUSED', '0') == '1'      # stamps of the fused-input-projection kernel
x = torch.randn(T, B, H, device=dev).to(torch.bfloat16)
wih = (torch.randn(8 * H, H, device=dev) * 0.05).to(torch.bfloat16)
for _ in range(3):
    if FUSED:
        y, ybf, gates, cs = _native.lstm_bidir_fwd_fused(x, wih, whh, lens)
    else:
        y, ybf, gates, cs = _native.lstm_bidir_fwd(gx, whh, lens)
    # (the backward the step runs: the fused-input-gradient variant is opt-in, ASR_LSTM_FUSED_BWD)
    dg = _native.lstm_bidir_bwd(dy, whhT, lens, gates, cs)
torch.cuda.synchronize()
R = int(os.environ.get('TILE_ROWS', '24'))    # batch rows per tile the host picked (24 at B=512)
v = y[0].reshape(B, 2, H)[0::R].reshape(-1, 2, H // 64, 64)[..., :6].reshape(-1, 6)
m = v.mean(0).tolist()
print('fwd B=%d T=%d cycles/step: %s  total %.0f' % (B, T, ' '.join('%.0f' % x for x in m), sum(m)))
print('     max over workgroups: %s' % ' '.join('%.0f' % x for x in v.max(0)[0].tolist()))
w = dg[0].reshape(B, 2, 4 * H)[0::R][..., :H].reshape(-1, 2, H // 64, 64)[..., :6].reshape(-1, 6).float() * 16
m = w.mean(0).tolist()
print('bwd B=%d T=%d cycles/step: %s  total %.0f' % (B, T, ' '.join('%.0f' % x for x in m), sum(m)))
