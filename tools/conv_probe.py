"""Time the DeepSpeech2 conv stack (fwd+bwd) in fp32 vs bf16 autocast (dev probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'gpurun_out', 'miopen_db_new'))
os.makedirs(os.environ['MIOPEN_USER_DB_PATH'], exist_ok=True)
import torch
from torch import nn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
conv = nn.Sequential(
    nn.Conv2d(1, 32, (7, 7), (1, 2), padding=(6, 0)), nn.BatchNorm2d(32), nn.Hardtanh(0, 20, inplace=True),
    nn.Conv2d(32, 32, (7, 7), (3, 1)), nn.BatchNorm2d(32), nn.Hardtanh(0, 20, inplace=True)).to(dev)
x = torch.randn(B, 1, 1000, 40, device=dev)


def run(mode, n=5):
    def step():
        if mode == 'bf16':
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = conv(x)
        elif mode == 'bf16_cl':
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = conv(x.contiguous(memory_format=torch.channels_last))
        elif mode.startswith('mixed'):
            import torch.nn.functional as F
            h = conv[2](conv[1](conv[0](x)))
            if mode == 'mixed_cl':
                hb = h.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
            else:
                hb = h.to(torch.bfloat16)
            z = F.conv2d(hb, conv[3].weight.to(torch.bfloat16), conv[3].bias.to(torch.bfloat16), (3, 1))
            y = conv[5](conv[4](z.float()))
        else:
            y = conv(x)
        y.float().sum().backward()
        return y
    t0 = time.time(); y = step(); torch.cuda.synchronize()
    print(mode, 'first call %.1f s' % (time.time() - t0), y.dtype, tuple(y.shape), flush=True)
    step(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(mode, '%.2f ms per fwd+bwd' % ((time.time() - t0) / n * 1e3), flush=True)


for m in sys.argv[2:] or ['fp32', 'bf16', 'bf16_cl']:
    run(m)
