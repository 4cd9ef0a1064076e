"""Per-phase s_memtime stamps of the conv2 forward / input-gradient item loops (dev probe).
Build the library with -DCONV_STAMPS first:
  make -C pytorch-asr_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DCONV_STAMPS"
then run this on the GPU box; the host entry points print the averaged phase cycles of one
workgroup (csrc/conv.hip, STAMP)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
from att_speech import _native
dev = torch.device('cuda:0')
B = 576
x = torch.randn(B, 32, 1006, 17, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(32, 32, 7, 7, device=dev) * 0.05
y = _native.conv7x7c32_fwd(x, w, 3)
dy = torch.randn_like(y)
_native.conv7x7c32_bwd_data(dy, w, 1006, 17, 3)
torch.cuda.synchronize()
os.environ['ASR_CONV_STAMPS'] = '1'
_native.conv7x7c32_fwd(x, w, 3)
_native.conv7x7c32_bwd_data(dy, w, 1006, 17, 3)
torch.cuda.synchronize()
