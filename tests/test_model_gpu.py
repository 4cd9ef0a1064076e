"""GPU parity of the module-level hot path: SpeechModel / FSTDecoder /
CTCDecoderAdvanced on the MI355X against a CPU composition of the same torch
modules plus the CPU oracle for the lattice arithmetic."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_modules import DEC_CDE, DEC_MONO, ENC, VOCAB, sample_batch  # noqa: E402


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def cpu_reference_loss(model_cpu, feats, lens, texts, llens, oracle, order=1,
                       denominator=False):
    """fp32 CPU evaluation of FSTDecoder.forward (advanced_decoder.py:454-534)
    with the oracle lattice; returns (loss, grads dict)."""
    from oracle import fst_oracle

    class OracleLattice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, lens_, mats):
            r = oracle.path_logsumexp(lp.detach().numpy(), np.asarray(lens_), mats)
            ctx.grads = torch.from_numpy(r['grad'])
            return torch.from_numpy(r['logZ'])

        @staticmethod
        def backward(ctx, g):
            return g[None, :, None] * ctx.grads, None, None

    dec = model_cpu.decoder
    S = dec.num_symbols
    enc, elens = model_cpu.encoder(feats, lens, None)
    logits = dec.fc(enc)
    if dec.normalize_by_dim is not None:
        logits = torch.log_softmax(logits, -1)
    mx = logits.max(-1, keepdim=True)[0].detach()
    mask = (torch.arange(logits.size(0))[:, None] < elens[None, :]).float()
    gg = fst_oracle.CTCGraphGen(S, order)
    num_m = gg.get_training_matrices_batch(texts.numpy(), llens.numpy())
    num = -OracleLattice.apply(logits - mx, elens, num_m)
    if denominator:
        den = -OracleLattice.apply(logits - mx, elens, gg.get_decoding_matrices())
    else:
        den = (mx.squeeze(-1) * mask).sum(0)
    loss = (num - den).sum()
    model_cpu.zero_grad()
    loss.backward()
    return float(loss), {k: p.grad.clone() for k, p in model_cpu.named_parameters()
                         if p.grad is not None}


def make_batch(B, T, S, L, order, seed, F=40, ch=1):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, F, ch, generator=g)
    lens = torch.tensor([T - 9 * b for b in range(B)], dtype=torch.int32)
    llens = torch.tensor([max(1, L - 2 * b) for b in range(B)], dtype=torch.int32)
    texts = torch.randint(1, S, (B, L), generator=g, dtype=torch.int32)
    if order == 2:
        prev = torch.cat([torch.zeros(B, 1, dtype=torch.int32), texts[:, :-1]], 1)
        texts = prev * S + texts
    return feats, lens, texts, llens


@pytest.mark.parametrize('cfg', ['mono_ctc', 'bigram_ctcg_cde', 'bigram_ctcg_cde_s49', 'mono_ctc_wsj'])
def test_speech_model_train_step_matches_cpu(oracle_lib, cfg):
    from att_speech.models import SpeechModel
    torch.manual_seed(7)
    B, T, L, F, ch = 4, 150, 10, 40, 1
    if cfg == 'mono_ctc':
        S, order, dec_cfg, vocab = 49, 1, DEC_MONO, VOCAB
    elif cfg == 'mono_ctc_wsj':
        # the shipped recipes' feature shape: 80 mel + energy with deltas, feat_dim [-1, 3, 81]
        # (egs/wsj/yamls/ctc.yaml:8-15) -> conv out 32 x 32 -> LSTM input 1024
        S, order, dec_cfg, vocab = 49, 1, DEC_MONO, VOCAB
        F, ch = 81, 3
    elif cfg == 'bigram_ctcg_cde':   # ctcg_bi_cde.yaml: global normalisation, NGramLinear embedder
        S, order, dec_cfg, vocab = 7, 2, DEC_CDE, VOCAB[:7]
    else:
        # the same recipe at its real size (egs/wsj/yamls/ctcg_bi_cde.yaml:69-74): 49 symbols,
        # C = 2401 classes, NGramLinear weight net, numerator + the 2401-state x 51-arc
        # grouped denominator together
        S, order, dec_cfg, vocab = 49, 2, DEC_CDE, VOCAB
        B, T, L = 2, 180, 8
    feats, lens, texts, llens = make_batch(B, T, S, L, order, 11, F, ch)
    sb = sample_batch(B=2, T=T, F=F, ch=ch)
    model = SpeechModel(ENC, dec_cfg, sb, S ** order, vocab)
    model.train()
    for mod in model.modules():    # BN in eval mode: batch statistics are not
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):   # part of the parity question
            mod.eval()
    ref = copy.deepcopy(model)
    want, want_g = cpu_reference_loss(ref, feats, lens, texts, llens, oracle_lib,
                                      order, denominator=(cfg != 'mono_ctc'))
    model.to(dev())
    out = model(feats.to(dev()), lens, None, texts, llens)
    assert set(out) == {'fst_loss', 'loss'}
    out['loss'].backward()
    got = float(out['loss'])
    # (a) whole model: the encoder GEMMs run with bf16 operands on the GPU
    # (BASELINE config: bf16), so end to end the match is at bf16 level
    assert abs(got - want) <= 2e-2 * abs(want), (got, want)
    for k, p in model.named_parameters():
        g, w = p.grad.cpu().flatten(), want_g[k].flatten()
        if float(w.norm()) > 1e-4:
            cos = float(torch.dot(g, w) / (g.norm() * w.norm() + 1e-20))
            assert cos > 0.98, (k, cos)
    # (b) the decoder + lattice arithmetic on IDENTICAL inputs: feed the GPU
    # encoder's output to the fp32 CPU evaluation -> north_star bound 1e-4
    with torch.no_grad():
        enc_g, elens = model.encoder(feats.to(dev()), lens, None)
    enc_c = enc_g.cpu().requires_grad_()
    ref.encoder.forward = lambda *a, **k: (enc_c, elens)
    want2, want_g2 = cpu_reference_loss(ref, feats, lens, texts, llens, oracle_lib,
                                        order, denominator=(cfg != 'mono_ctc'))
    enc_in = enc_g.clone().requires_grad_()
    model.zero_grad()
    out3 = model.decoder(enc_in, elens, texts, llens)
    out3['loss'].backward()
    assert abs(float(out3['loss']) - want2) <= 1e-4 * abs(want2), (float(out3['loss']), want2)
    for k, p in model.decoder.named_parameters():
        g, w = p.grad.cpu(), want_g2['decoder.' + k]
        scale = max(float(w.abs().max()), 1e-3)
        # relative to the tensor's largest gradient, plus an absolute floor for
        # gradients that cancel analytically (e.g. a global bias under CTC-G: numerator and
        # denominator posteriors both sum to the frame count); the floor is the fp32
        # summation noise of that cancellation, which grows with the number of classes
        floor = 1e-5 * max(1.0, (S ** order) / 49.0) ** 0.5
        assert float((g - w).abs().max()) <= 2e-3 * scale + floor, (k, float((g - w).abs().max()))
    ge = enc_c.grad
    assert float((enc_in.grad.cpu() - ge).abs().max()) <= 2e-3 * float(ge.abs().max()) + 1e-6
    # graph matrices handed over by the data pipeline (first-batch self check,
    # advanced_decoder.py:460-468) give the same loss
    gm = model.decoder.graph_generator.get_training_matrices_batch(texts, llens)
    out2 = model(feats.to(dev()), lens, None, texts, llens, graph_matrices=gm)
    assert abs(float(out2['loss']) - got) <= 1e-6 * abs(got)


def test_fst_decoder_decode_is_bit_exact(oracle_lib):
    from att_speech.models import SpeechModel
    from oracle import fst_oracle
    torch.manual_seed(3)
    for S, order, dec_cfg, vocab in [(49, 1, DEC_MONO, VOCAB), (7, 2, DEC_CDE, VOCAB[:7])]:
        B, T = 5, 120
        feats, lens, texts, llens = make_batch(B, T, S, 8, order, 5)
        model = SpeechModel(ENC, dec_cfg, sample_batch(B=2, T=T), S ** order, vocab).to(dev())
        model.eval()
        out = model.decode(feats.to(dev()), lens, None, texts, llens)
        logits = out['logits'].detach().cpu().numpy()
        elens = ((lens + 2) // 3).numpy()
        gg = fst_oracle.CTCGraphGen(S, order)
        _, best = oracle_lib.path_forward(logits, elens, gg.get_decoding_matrices(),
                                          viterbi=True)
        want = [fst_oracle.read_out_olabels(gg.decoding_fst, best[:elens[b], b])
                for b in range(B)]
        assert out['decoded'] == want                      # bit-exact label indices
        assert 'loss' in out and set(out['loss']) == {'fst_loss', 'loss'}


def test_ctc_decoder_advanced_matches_torch_ctc():
    from att_speech.modules.decoders import CTCDecoderAdvanced
    torch.manual_seed(5)
    S, B, T, H = 49, 6, 40, 32
    enc = torch.randn(T, B, H)
    elens = torch.tensor([40, 38, 33, 30, 21, 20], dtype=torch.int32)
    llens = torch.tensor([9, 7, 7, 5, 3, 1], dtype=torch.int32)
    texts = torch.randint(1, S, (B, 9), dtype=torch.int32)
    dec = CTCDecoderAdvanced({'features': torch.zeros(T, 2, H)}, S, vocabulary=VOCAB)
    ref = copy.deepcopy(dec)
    dec.to(dev())
    x = enc.to(dev()).requires_grad_()
    out = dec(x, elens, texts, llens)
    out['loss'].backward()
    xr = enc.clone().requires_grad_()
    lp = torch.log_softmax(ref.fc(xr), -1)
    cat = torch.cat([texts[b, :llens[b]] for b in range(B)]).long()
    want = torch.nn.functional.ctc_loss(lp, cat, elens.long(), llens.long(),
                                        reduction='mean')          # ctc_losses.py:62
    want.backward()
    assert abs(float(out['loss']) - float(want)) <= 1e-4 * abs(float(want))
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), atol=2e-5)
    # greedy decode: per-frame arg-max, bug-compatible collapse
    with torch.no_grad():
        res = dec.decode(enc.to(dev()), elens)
        logits = torch.log_softmax(ref.fc(enc), -1)
    frames = logits.argmax(-1).transpose(0, 1)
    np.testing.assert_array_equal(res['decoded_frames'].numpy(), frames.numpy())
    assert res['decoded'] == ref.process_sequences(frames, elens)


def test_frame_projection_backward_matches_linear():
    """LutLinear's GPU backward (chunked dW / bias reductions) against F.linear's."""
    from att_speech.modules.decoders.advanced_decoder import _FrameProjection
    torch.manual_seed(1)
    d = dev()
    x = torch.randn(96, 64, 40, device=d)
    w = torch.randn(49, 40, device=d)
    b = torch.randn(49, device=d)
    dy = torch.randn(96, 64, 49, device=d)
    ref = [t.clone().requires_grad_() for t in (x, w, b)]
    torch.nn.functional.linear(*ref).backward(dy)
    got = [t.clone().requires_grad_() for t in (x, w, b)]
    y = _FrameProjection.apply(*got)
    y.backward(dy)
    assert torch.equal(y, torch.nn.functional.linear(x, w, b))
    for a, r in zip(got, ref):
        torch.testing.assert_close(a.grad, r.grad, rtol=1e-4, atol=1e-3)


def test_train_step_with_bucket_and_hooks_on_the_device(oracle_lib):
    """SURVEY.md §8f N1 on the GPU: SpeechModel + FlatGradBucket + GradientClipping +
    PolyakDecay through dp.train_step (world size 1).  The clipped flat bucket equals
    clip_grad_norm_ on an identical replica without a bucket, a step over skip_step_norm
    leaves the weights alone, otherwise Adam moves them and the Polyak average follows."""
    from att_speech.dp import FlatGradBucket, train_step
    from att_speech.models import SpeechModel
    from att_speech.modules.hooks import GradientClipping, PolyakDecay
    torch.manual_seed(21)
    B, T, L = 4, 120, 8
    feats, lens, texts, llens = make_batch(B, T, 49, L, 1, 13)
    model = SpeechModel(ENC, DEC_MONO, sample_batch(B=2, T=T), 49, VOCAB).to(dev())
    twin = copy.deepcopy(model)
    args = ((feats.to(dev()), lens, None, texts, llens), {})
    # (1) clipping on the flat bucket == clip_grad_norm_ on separate gradients
    bucket = FlatGradBucket(model.parameters())
    hook = GradientClipping(clip_norm=0.5, skip_step_norm=1e12)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    out, skipped = train_step(model, opt, args, hooks=[hook], bucket=bucket)
    assert not skipped and hook.gstats.clips == 1
    assert abs(float(bucket.flat.norm()) - 0.5) < 1e-3
    twin(*args[0])['loss'].backward()
    total = float(torch.nn.utils.clip_grad_norm_(twin.parameters(), 0.5))
    assert abs(hook.gstats.norms[0] - total) <= 2e-3 * total
    for (k, p), q in zip(model.named_parameters(), twin.parameters()):
        scale = max(float(q.grad.abs().max()), 1e-6)
        assert float((p.grad - q.grad).abs().max()) <= 2e-2 * scale, k     # bf16 encoder, two runs
    # (2) skip: weights and optimizer state untouched; (3) a normal step moves them
    before = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    polyak = PolyakDecay(decay_rates=[0.5])
    polyak.pre_run(model, opt)
    skipper = GradientClipping(clip_norm=1e12, skip_step_norm=1e-9)
    out, skipped = train_step(model, opt, args, hooks=[skipper, polyak], bucket=bucket)
    assert skipped and all(torch.equal(p, before[k]) for k, p in model.named_parameters())
    out, skipped = train_step(model, opt, args,
                              hooks=[GradientClipping(clip_norm=1e12, skip_step_norm=1e12), polyak],
                              bucket=bucket)
    assert not skipped
    w_new, w_old = model.state_dict()['decoder.fc.0.module.0.weight'], before['decoder.fc.0.module.0.weight']
    assert not torch.equal(w_new, w_old)
    avg = getattr(model, PolyakDecay.dict_name(0.5))['decoder.fc.0.module.0.weight']
    # two post_optimizer_step calls: avg <- old (no step), then avg <- 0.5 old + 0.5 new
    torch.testing.assert_close(avg, 0.5 * w_old + 0.5 * w_new, rtol=1e-5, atol=1e-7)
    losses = [float(train_step(model, opt, args, hooks=[polyak], bucket=bucket)[0]['loss'])
              for _ in range(5)]
    assert losses[-1] < losses[0]
