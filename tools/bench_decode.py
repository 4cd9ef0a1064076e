"""Decode-side measurements (development tool; bench.py is the contract benchmark):
config 1-3 greedy / Viterbi decode through FSTDecoder.decode and config 4, the TCN
attention decoder with plain BeamSearch (egs/wsj/yamls/lattice_decoding/tcn.yaml
shapes: tcn_hidden_size 384, dilations [1, 2], 2 layers per block, beam_size from
--beam).  Synthetic 40-dim x 1000-frame features, random weights; prints utterances/s
and input frames/s."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'pytorch-asr_amd', 'miopen_db'))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-asr_amd'))
sys.path.insert(0, ROOT)

import bench                                   # noqa: E402  (model_config, synthetic_batch)
from att_speech.models import SpeechModel       # noqa: E402

S = 49


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--beam', type=int, default=10)
    ap.add_argument('--iters', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, T = a.batch, a.frames
    feats, lens, texts, llens = bench.synthetic_batch(B, T, 0, 1)

    def sample():      # SpeechModel replaces sample_batch['features'] by the encoded probe (models.py:29-30)
        return {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    enc_cfg, dec_cfg = bench.model_config(1, None)
    torch.manual_seed(0)
    with torch.no_grad():
        # ---- FSTDecoder: encoder + logits + Viterbi read-out (advanced_decoder.py:519-593)
        model = SpeechModel(enc_cfg, dec_cfg, sample(), S, [str(i) for i in range(S)]).to(dev).eval()
        f = feats.to(dev)
        dt = timeit(lambda: model.decode(f, lens, None, texts, llens), a.iters)
        print('FSTDecoder Viterbi decode  B=%d: %.1f ms/batch  %.0f utt/s  %.2f M frames/s'
              % (B, dt * 1e3, B / dt, B * T / dt / 1e6))
        # ---- TCN attention decoder + BeamSearch on the same encoder
        tcn_cfg = dict(class_name='att_speech.modules.tcn.AttentionDecoderTCN',
                       att_hidden_size=64, beam_size=a.beam, dilation_sizes=[1, 2], dropout_p=0.3,
                       kernel_size=3, length_normalization=0.6, tcn_hidden_size=384,
                       tcn_layers_per_block=2)
        enc_cfg2, _ = bench.model_config(1, None)
        model2 = SpeechModel(enc_cfg2, tcn_cfg, sample(), S, [str(i) for i in range(S)]).to(dev).eval()
        model2.decoder.TRANSCRIPTION_LEN_GUARD = 120
        dt = timeit(lambda: model2.decode(f, lens, None, texts, llens), max(1, a.iters - 1))
        print('TCN + BeamSearch(beam=%d) decode B=%d: %.1f ms/batch  %.1f utt/s  %.3f M frames/s'
              % (a.beam, B, dt * 1e3, B / dt, B * T / dt / 1e6))


if __name__ == '__main__':
    main()
