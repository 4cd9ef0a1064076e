import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
import torch, bench
from att_speech.models import SpeechModel
B = int(sys.argv[1]); T = 1000
dev = torch.device('cuda:0')
def log(*a):
    print(*a, flush=True)
t0 = time.time()
feats, lens, texts, llens = bench.synthetic_batch(B, T, 0, 1)
enc_cfg, dec_cfg = bench.model_config(1)
torch.manual_seed(1234)
sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
model = SpeechModel(enc_cfg, dec_cfg, sb, 49, [str(i) for i in range(49)]).to(dev)
log('model built', time.time() - t0)
gm = model.decoder.graph_generator.get_training_matrices_batch(texts, llens)
log('graphs', time.time() - t0)
fd = feats.to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for i in range(4):
    torch.cuda.synchronize(); t1 = time.time()
    opt.zero_grad()
    enc, el = model.encoder(fd, lens, None)
    torch.cuda.synchronize(); t2 = time.time()
    out = model.decoder(enc, el, texts, llens, graph_matrices=gm)
    torch.cuda.synchronize(); t3 = time.time()
    out['loss'].backward()
    torch.cuda.synchronize(); t4 = time.time()
    opt.step()
    torch.cuda.synchronize(); t5 = time.time()
    log('step %d: enc %.1f ms  dec %.1f ms  bwd %.1f ms  adam %.1f ms  loss %.3f' % (
        i, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, float(out['loss'])))
