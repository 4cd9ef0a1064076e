#!/usr/bin/env python
"""bench.py — CTC train-step frames/s on MI355X (BASELINE.json metric).

One "step" = forward + backward + Adam of SpeechModel(DeepSpeech2 encoder +
FSTDecoder, mono-char CTC: egs/wsj/yamls/ctc.yaml shapes) on one synthetic batch
of B utterances x 1000 fbank frames x 40 dims per GPU; frames/s is the reference's
own throughput definition, sum(feature_lens) / step_time (trainer.py:292-298).
N > 1: one process per GPU (torch.distributed.run), utterance-sharded data
parallelism, one RCCL all-reduce of the flat gradient bucket per step.

`python bench.py --gpus N` without WORLD_SIZE in the environment starts the N ranks
itself (a `torch.distributed.run` child process, before anything touches the GPU),
relays rank 0's JSON line and exits with the child's code.

Prints ONE JSON line on rank 0 (see the contract in the task statement), with
  roofline        — the lattice forward-backward scan kernel of the step, HBM-bound,
                    timed live with events on the stream it is launched on;
  roofline_bichar — the scan on the bi-char numerator (C = 2401, B = 512), launched alone
                    after the timed region (N = 1 only); `frac` on algorithmic bytes,
                    `frac_traffic` on the HBM bytes the counters saw;
  extra_workloads — the training step of BASELINE configs 3 and 5 beside the headline:
                    bi-char CTC (ctc_bi) and bi-char CTC-G + CDE (ctcg_bi_cde), a few timed
                    steps each after the headline region (with N > 1: the bi-char step, the
                    model of config 5, on all ranks);
  roofline_mfma   — dense flops of the step / step time / 2.5 PFLOP/s (bf16 dense peak);
  decode          — BASELINE config 4 beside the headline: utterances/s of the TCN attention
                    decoder with beam 10 (N = 1 only);
  loss_delta      — |loss_gpu - loss_cpu| / |loss_cpu| of one small batch, product
                    model on the GPU vs the fp32 CPU composition with the same weights
                    (end to end, and decoder + lattice on identical encoder output);
  cpu_baseline    — the same step on the host cores (torch-CPU encoder + the CPU
                    oracle for the lattice), bounded sample, N = 1 only.
"""
import argparse
import contextlib
import copy
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# pinned MIOpen solver choice for convolution shapes outside csrc/conv.hip (none in the
# bench workloads; see pytorch-asr_amd/miopen_db/README.md); must be set before MIOpen loads
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'pytorch-asr_amd', 'miopen_db'))


def pin_gemm_selection(local_rank):
    """Library GEMM solutions measured fastest for this step's bf16 GEMM shapes (PyTorch
    TunableOp results committed under pytorch-asr_amd/tunableop/, tuning OFF): the same
    kind of pin as the MIOpen find-db above.  TunableOp reads `<name><device>.csv`, so
    every rank gets its own copy (in its own temporary directory).  Any PYTORCH_TUNABLEOP_* setting of the caller wins."""
    src = os.path.join(ROOT, 'pytorch-asr_amd', 'tunableop', 'gfx950.csv')
    if any(k.startswith('PYTORCH_TUNABLEOP_') for k in os.environ) or not os.path.exists(src):
        return
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix='asr_tunableop_')
    for dev_ordinal in sorted({0, local_rank}):      # ordinal 0 too: launchers that mask devices per rank
        shutil.copy(src, os.path.join(d, 'results%d.csv' % dev_ordinal))
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='0',
                      PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, 'results.csv'))
for p in (ROOT, os.path.join(ROOT, 'pytorch-asr_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

_STDOUT = sys.stdout
HBM_PEAK_GBPS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0               # MI355X_MICROARCH.md: bf16 MFMA ~2.5 PF dense
S = 49                                  # egs/wsj/vocabulary.txt


WORKLOADS = {          # BASELINE.json configs -> (context order, yaml)
    'ctc': (1, 'ctc'), 'ctc_bi': (2, 'ctc_bi'), 'ctcg_bi_cde': (2, 'ctcg_bi_cde')}


def model_config(order=1, workload=None):
    enc = dict(class_name='att_speech.modules.encoders.DeepSpeech2',
               conv_kernel_sizes=[[7, 7], [7, 7]], conv_strides=[[1, 2], [3, 1]],
               rnn_hidden_size=320, rnn_nb_layers=4, rnn_normalization='none')
    dec = dict(class_name='att_speech.modules.decoders.advanced_decoder.FSTDecoder',
               denominator_red='none', normalize_by_dim=0,
               graph_generator=dict(class_name='CTCGraphGen', context_order=order))
    if workload == 'ctcg_bi_cde':      # egs/wsj/yamls/ctcg_bi_cde.yaml: global normalisation
        dec = dict(class_name='att_speech.modules.decoders.advanced_decoder.FSTDecoder',
                   graph_generator=dict(class_name='CTCGraphGen', context_order=2),
                   embedder='NGramLinear',
                   embedder_kwargs=dict(bias_only_for_dim=1, num_layers=3,
                                        embedding_combination_method='concat',
                                        tied_embeddings=False))
    return enc, dec


FEATS = (40, 1)      # (feature bins, channels); --features wsj: (81, 3), the WSJ recipes' real shape


def synthetic_batch(B, T, rank, order=1):
    """features N(0,1) [B,T,40,1]; saturating batch = all-T
    lengths (B > 16), YAML batch = T - 8b; labels uniform in [2,48],
    L_b = 100 - 2 (b mod 16)."""
    g = torch.Generator().manual_seed(1234 + rank)
    feats = torch.randn(B, T, FEATS[0], FEATS[1], generator=g)
    lens = torch.tensor([T if B > 16 else T - 8 * b for b in range(B)], dtype=torch.int32)
    llens = torch.tensor([100 - 2 * (b % 16) for b in range(B)], dtype=torch.int32)
    texts = torch.randint(2, S, (B, 100), generator=g, dtype=torch.int32)
    if order == 2:      # bigram ids prev*S+cur (egs/wsj/data.py:146-161)
        prev = torch.cat([torch.zeros(B, 1, dtype=torch.int32), texts[:, :-1]], 1)
        texts = prev * S + texts
    for b in range(B):
        texts[b, llens[b]:] = 0
    return feats, lens, texts, llens


def lattice_algorithmic_bytes(enc_lens, C, label_lens, n_arcs):
    """sum_b 4*T'_b*(3C + 2N_b) + 2*E_b*12 + 4*N_b."""
    tl = np.asarray(enc_lens, np.int64)
    ns = 2 * np.asarray(label_lens, np.int64) + 1
    return int((4 * tl * (3 * C + 2 * ns) + 24 * np.asarray(n_arcs, np.int64) + 4 * ns).sum())


def step_flops(B, T, C, in_feats=None, H=320, layers=4):
    """Dense multiply-add flops of one training step (SURVEY.md §8d 'Algorithmic
    flops'): both convolutions, the 4 BiLSTM layers (input + recurrent products)
    and the class projection; training = 3 x forward (fwd, dgrad, wgrad)."""
    in_feats, cin = (FEATS[0] if in_feats is None else in_feats), FEATS[1]
    t1 = T + 2 * 6 - 7 + 1                      # conv1: k 7x7, stride (1,2), pad (6,0)
    f1 = (in_feats - 7) // 2 + 1
    t2 = (t1 - 7) // 3 + 1                      # conv2: k 7x7, stride (3,1)
    f2 = f1 - 7 + 1
    fwd = 2.0 * B * 32 * t1 * f1 * 49 * cin + 2.0 * B * 32 * t2 * f2 * 32 * 49
    rnn_in = 32 * f2
    for l in range(layers):
        fwd += 2 * 2.0 * t2 * B * 4 * H * ((rnn_in if l == 0 else H) + H)
    fwd += 2.0 * t2 * B * H * C
    return 3.0 * fwd


class _OracleLattice(torch.autograd.Function):
    """PathLogSumExp evaluated by oracle/lattice_oracle.c (the checker)."""

    @staticmethod
    def forward(ctx, lp, lens, mats):
        from oracle import oracle
        r = oracle.path_logsumexp(lp.detach().numpy(), lens.numpy(),
                                  [m.numpy() for m in mats])
        ctx.grads = torch.from_numpy(r['grad'])
        return torch.from_numpy(r['logZ'])

    @staticmethod
    def backward(ctx, g):
        return g[None, :, None] * ctx.grads, None, None


def _cpu_loss(model, gg, enc, elens, texts, llens):
    """FSTDecoder.forward (advanced_decoder.py:454-534) in fp32 torch-CPU ops with
    the oracle lattice, given the encoder output."""
    logits = model.decoder.fc(enc)
    lp = torch.log_softmax(logits, -1)
    mx = lp.max(-1, keepdim=True)[0].detach()
    mask = (torch.arange(lp.size(0))[:, None] < elens[None, :]).float()
    mats = gg.get_training_matrices_batch(texts, llens)
    num = -_OracleLattice.apply(lp - mx, elens, mats)
    return (num - (mx.squeeze(-1) * mask).sum(0)).sum()


def cpu_baseline(T, order, dev=None, state_dict=None, seconds_budget=15.0):
    """The same training step on the host: torch-CPU encoder/projection +
    oracle/lattice_oracle.c for the lattice (kind 'port'), small batch.  With
    `dev`: also the loss delta of that batch between the product model on the
    GPU and this fp32 CPU composition, both carrying `state_dict` (the weights the
    timed steps trained: peaked outputs; at random init the loss is insensitive).
    Returns (cpu_baseline dict, loss_delta dict or None)."""
    from att_speech.models import SpeechModel
    from att_speech import fst_utils

    # the box's CPU share for one GPU is 16 cores; os.cpu_count() reports the
    # whole host and oversubscribing it is pathologically slow
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    B = 4
    feats, lens, texts, llens = synthetic_batch(B, T, 0, order)
    enc_cfg, dec_cfg = model_config(order)
    torch.manual_seed(1234)
    sb = {'features': feats.clone(), 'features_lengths': lens.clone(), 'spkids': None}
    model = SpeechModel(enc_cfg, dec_cfg, sb, S ** order, [str(i) for i in range(S)])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gg = fst_utils.CTCGraphGen(context_order=order, num_symbols=S)

    delta = None
    if dev is not None:
        if state_dict is not None:
            model.load_state_dict({k: v.detach().float().cpu() for k, v in state_dict.items()})
        gpu_model = copy.deepcopy(model).to(dev)
        with torch.no_grad():
            enc_c, elens = model.encoder(feats, lens, None)
            loss_cpu = float(_cpu_loss(model, gg, enc_c, elens, texts, llens))
            loss_gpu = float(gpu_model(feats.to(dev), lens, None, texts, llens)['loss'])
            enc_g, elens_g = gpu_model.encoder(feats.to(dev), lens, None)
            loss_dec_gpu = float(gpu_model.decoder(enc_g, elens_g, texts, llens)['loss'])
            loss_dec_cpu = float(_cpu_loss(model, gg, enc_g.float().cpu(), elens, texts, llens))
        delta = {
            'end_to_end_rel': abs(loss_gpu - loss_cpu) / abs(loss_cpu),
            'decoder_lattice_rel': abs(loss_dec_gpu - loss_dec_cpu) / abs(loss_dec_cpu),
            'loss_gpu': loss_gpu, 'loss_cpu': loss_cpu,
            'batch': '%d x %d frames, the weights after the timed steps on both sides; end_to_end = bf16-operand encoder on '
                     'the GPU vs fp32 torch-CPU encoder; decoder_lattice = projection + '
                     'log-softmax + lattice on the SAME (GPU) encoder output, fp32 both '
                     'sides (north_star bound 1e-4)' % (B, T)}
        del gpu_model

    def step():
        opt.zero_grad()
        enc, elens = model.encoder(feats, lens, None)
        loss = _cpu_loss(model, gg, enc, elens, texts, llens)
        loss.backward()
        opt.step()
        return float(loss)

    step()                                   # warm-up
    t0, n = time.time(), 0
    while n < 2 or (time.time() - t0 < seconds_budget and n < 50):
        step()
        n += 1
    dt = (time.time() - t0) / n
    base = dict(value=float(lens.sum()) / dt, unit='frames/s', cores=cores, kind='port',
                sample='%d steps of the same train step at B=%d x %d frames on the host: '
                       'torch-CPU conv/BiLSTM/projection on %d threads (the part that bounds '
                       'it) + oracle/lattice_oracle.c lattice on 1 thread'
                       % (n, B, T, cores))
    return base, delta


def bichar_numerator_roofline(dev, B=512, Tp=334, iters=10, traffic=None):
    """The alpha/beta scan alone on the bi-char numerator (ctc_bi shape: C = 2401,
    L_b = 100 - 2 (b mod 16), all utterances Tp frames): average launch time from
    events on the launch stream / SURVEY.md §8d algorithmic bytes."""
    from att_speech import _native, fst_utils
    C = S * S
    _, _, texts, llens = synthetic_batch(B, 3 * Tp - 2, 0, 2)
    gg = fst_utils.CTCGraphGen(context_order=2, num_symbols=S)
    mats = gg.get_training_matrices_batch(texts, llens)
    n_arcs = (mats[2] > -1e19).sum((1, 2)).numpy()
    g = _native.Graph(mats, dev)
    gen = torch.Generator(device=dev).manual_seed(4321)
    lp = _native.log_softmax_fwd(torch.randn(Tp, B, C, device=dev, generator=gen), C)
    tl = torch.full((B,), Tp, dtype=torch.int32, device=dev)
    for _ in range(2):
        _native.lattice_fwbw(lp, tl, g)
    ev = []
    _native.EVENT_HOOK = ev
    try:
        for _ in range(iters):
            _native.lattice_fwbw(lp, tl, g)
    finally:
        _native.EVENT_HOOK = None
    torch.cuda.synchronize()
    ms = float(np.mean([s.elapsed_time(e) for (s, e) in ev]))
    alg = lattice_algorithmic_bytes(np.full(B, Tp), C, llens.numpy(), n_arcs)
    ach = alg / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': 'lattice_fwbw (alpha/beta scan), bi-char numerator '
                                      'C=2401 B=%d T\'=%d' % (B, Tp),
            'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
            'traffic': traffic,
            'frac_traffic': (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            'algorithmic_bytes_per_launch': alg, 'avg_launch_ms': ms}


def tcn_decode_rate(dev, B=64, T=1000, beam=10, steps=120):
    """BASELINE config 4: encoder + AttentionDecoderTCN (lattice_decoding/tcn.yaml dimensions)
    with beam search on one GPU; random weights rarely emit EOS, so the label-step budget is
    fixed.  Utterances per second over whole `decode` calls."""
    from att_speech.models import SpeechModel
    feats, lens, texts, llens = synthetic_batch(B, T, 0, 1)
    enc_cfg, _ = model_config(1, None)
    dec_cfg = dict(class_name='att_speech.modules.tcn.AttentionDecoderTCN', att_hidden_size=64,
                   beam_size=beam, dilation_sizes=[1, 2], dropout_p=0.3, kernel_size=3,
                   length_normalization=0.6, tcn_hidden_size=384, tcn_layers_per_block=2)
    torch.manual_seed(0)
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    model = SpeechModel(enc_cfg, dec_cfg, sb, S, [str(i) for i in range(S)]).to(dev).eval()
    model.decoder.TRANSCRIPTION_LEN_GUARD = steps
    f = feats.to(dev)
    with torch.no_grad():
        model.decode(f, lens, None)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 3
        for _ in range(n):
            model.decode(f, lens, None)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    return {'metric': 'stage-2 decode: DeepSpeech2 encoder + TCN/local-attention decoder, beam search',
            'value': B / dt, 'unit': 'utt/s', 'ms_per_batch': dt * 1e3, 'batch': B, 'beam': beam,
            'label_steps': steps, 'frames': T,
            'workload': 'lattice_decoding/tcn.yaml dimensions, random weights, fixed label-step budget'}


def time_extra_workload(workload, B, T, steps, warmup, dev, rank, world, hooks_on=True, fused_on=True):
    """A few timed training steps of another BASELINE config (same step definition, same
    hooks, same barrier / max-over-ranks timing as the headline) -> dict for `extra_workloads`."""
    from att_speech.dp import FlatGradBucket, broadcast_parameters, train_step
    from att_speech.modules.hooks import GradientClipping, PolyakDecay
    from att_speech.models import SpeechModel
    order = WORKLOADS[workload][0]
    C = S ** order
    feats, lens, texts, llens = synthetic_batch(B, T, rank, order)
    enc_cfg, dec_cfg = model_config(order, workload)
    torch.manual_seed(1234)
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    model = SpeechModel(enc_cfg, dec_cfg, sb, C, [str(i) for i in range(S)]).to(dev)
    broadcast_parameters(model)
    bucket = FlatGradBucket(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    feats_d = feats.to(dev)
    hooks = []
    if hooks_on:
        scale = B * world / 16.0
        hooks = [GradientClipping(clip_norm=10000.0 * scale, skip_step_norm=100000.0 * scale),
                 PolyakDecay(decay_rates=[0.9998])]
        for h in hooks:
            h.pre_run(model, opt)
    fused = None
    if fused_on and hooks_on:
        from att_speech.fused_step import FusedClipAdam
        fused = FusedClipAdam.from_optimizer(opt, bucket, hooks[0])
    skipped = []

    def step(record):
        with contextlib.redirect_stdout(sys.stderr):
            out, skip = train_step(model, opt, ((feats_d, lens, None, texts, llens), {}),
                                   hooks=hooks, bucket=bucket, fused=fused)
        if record:
            skipped.append(bool(skip))
        return out['loss']

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step(False)
    fence()
    t0 = time.time()
    for _ in range(steps):
        loss = step(True)
    fence()
    if fused is not None:        # the device's decisions of the timed steps
        skipped = [a or rec[2] for a, rec in zip(skipped, fused.drain()[-steps:])]
    t = torch.tensor([time.time() - t0], dtype=torch.float64, device=dev)
    frames = torch.tensor([float(lens.sum())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    res = {'workload': '%s (egs/wsj/yamls/%s.yaml shapes), fwd+bwd+hooks+Adam, synthetic 40-dim x %d-frame fbank'
                       % (workload, WORKLOADS[workload][1], T),
           'batch_per_gpu': B, 'classes': C, 'steps': steps, 'warmup': warmup,
           'ms_per_step': dt / steps * 1e3, 'frames_per_s': float(frames.item()) * steps / dt,
           'optimizer_steps': len(skipped) - sum(skipped), 'final_loss': float(loss.detach())}
    del model, bucket, opt, feats_d
    torch.cuda.empty_cache()
    return res


def self_launch(a, argv):
    """`python bench.py --gpus N` with no rank environment: start the N ranks as a
    child `torch.distributed.run` (this process has not touched the GPU and never
    will), relay rank 0's JSON line, exit with the child's return code."""
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, universal_newlines=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith('{') and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, file=_STDOUT, flush=True)
    elif rc == 0:
        rc = 1
    sys.exit(rc)


def dry_run(a, world, rank):
    """Launcher rehearsal without a GPU (tests/test_bench_helpers.py): gloo ranks,
    the same barrier / max-over-ranks timing / single JSON line, no model."""
    if world > 1:
        dist.init_process_group('gloo')
    t0 = time.time()
    for _ in range(a.steps):
        time.sleep(0.001)
    t = torch.tensor([time.time() - t0], dtype=torch.float64)
    units = torch.tensor([float(a.batch * a.frames)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({'metric': 'launcher dry run', 'dry_run': True, 'n_gpus': world,
                          'steps': a.steps, 'warmup': a.warmup,
                          'value': float(units) * a.steps / float(t)}), file=_STDOUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


def mono_lattice_kernel(B):
    """The kernel att_speech._native.lattice_fwbw launches for a mono-char CTC batch of B
    utterances (band kernel up to _BAND_MAX_BATCH utterances, the state-labelled one above)."""
    from att_speech import _native
    return 'lattice_fwbw_band_kernel' if B <= _native._BAND_MAX_BATCH else 'lattice_fwbw_sl_kernel<3, 8, 1>'


def pmc_traffic(order, B, T, kernel_tag=None, grid=None):
    """HBM bytes per lattice launch from the PMC passes committed under profiles/
    (r03_pmc_step_fetch_write.json: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    runs of bench.py summarised by tools/pmc_summary.py; FETCH_SIZE counts half of a
    4 B/lane coalesced stream on gfx950 - calibrated on log_softmax_fwd - so
    traffic = 2*FETCH + WRITE).  bench.py cannot collect counters itself; the figure
    applies to the measured shape only (T = 1000 frames, the batch recorded in the file;
    the bi-char numerator launch of `roofline_bichar` is part of the same runs)."""
    if T != 1000:
        return None
    k = kernel_tag or mono_lattice_kernel(B) if order == 1 else kernel_tag or 'lattice_fwbw_sl_kernel<3, 8, 0>'
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r03_pmc_step_fetch_write.json')))
        if pmc.get('batch') != B:
            return None
        e = next(v for n, v in pmc['kernels'].items() if k in n)
        if 'by_grid' in e:          # several problem sizes in the run: the launch with `grid` work items
            # (the bi-char probe: 512 utterances x 512 threads), else the largest one
            key = str(grid) if grid is not None and str(grid) in e['by_grid'] else max(e['by_grid'], key=int)
            e = e['by_grid'][key]
        return (2 * e['fetch_KB'] + e['write_KB']) * 1024.0
    except (OSError, KeyError, ValueError, StopIteration, TypeError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=768,
                    help='utterances per GPU (768 = 24 batch tiles of 32 rows: the persistent LSTM '
                         'grid of 48 teams x 5 workgroups exactly, no padding rows in its MFMA '
                         'tiles; 576 = the same grid with 24-row tiles, 7 %% fewer frames/s)')
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--order', type=int, default=1, help='1 mono-char CTC, 2 bi-char CTC')
    ap.add_argument('--features', default='40x1', choices=['40x1', 'wsj'],
                    help="40x1: the BASELINE metric's synthetic 40-dim fbank (default); wsj: 81 mel bins x 3 "
                         "channels, the shape of the WSJ recipes' real features (egs/wsj/yamls/ctc.yaml:8-15)")
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help='BASELINE config; default ctc (mono-char, the headline metric)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hooks', action='store_true',
                    help='leave out GradientClipping / PolyakDecay (recipe hooks)')
    ap.add_argument('--host-step', action='store_true',
                    help='clip / skip decision and Adam on the host side (GradientClipping hook + '
                         'torch.optim.Adam, one read-back per step) instead of att_speech.fused_step')
    ap.add_argument('--sync-bn', action='store_true',
                    help='N > 1: BatchNorm statistics over all replicas (att_speech.dp.enable_sync_batchnorm), '
                         'as in the single-process reference, instead of per replica')
    ap.add_argument('--no-extra', action='store_true',
                    help='leave out the bi-char numerator roofline launch and the decode rate')
    ap.add_argument('--dry-run-launcher', action='store_true', help=argparse.SUPPRESS)
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        self_launch(a, sys.argv[1:])             # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if a.dry_run_launcher:
        return dry_run(a, world, rank)
    pin_gemm_selection(local_rank)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    from att_speech import _native, fst_utils
    from att_speech.dp import FlatGradBucket, broadcast_parameters, train_step
    from att_speech.modules.hooks import GradientClipping, PolyakDecay
    from att_speech.models import SpeechModel

    if a.workload:
        a.order = WORKLOADS[a.workload][0]
    global FEATS
    if a.features == 'wsj':
        FEATS = (81, 3)
    B, T, order = a.batch, a.frames, a.order
    C = S ** order
    feats, lens, texts, llens = synthetic_batch(B, T, rank, order)
    enc_cfg, dec_cfg = model_config(order, a.workload)
    torch.manual_seed(1234)
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    model = SpeechModel(enc_cfg, dec_cfg, sb, C, [str(i) for i in range(S)]).to(dev)
    broadcast_parameters(model)
    if a.sync_bn:
        from att_speech.dp import enable_sync_batchnorm
        enable_sync_batchnorm(True)
    bucket = FlatGradBucket(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    feats_d = feats.to(dev)                          # inputs resident in HBM
    # The training lattices are built ON the device every step from the labels
    # (asr_ctc_graph_build; the reference builds them per batch with OpenFst in
    # its data workers, kaldi_dataset.py:230-232, and copies 8 padded tensors).
    # The host builder is used once, outside the timed region, to count arcs
    # for the algorithmic-bytes formula.
    n_arcs = (model.decoder.graph_generator.get_training_matrices_batch(texts, llens)[2]
              > -1e19).sum((1, 2)).numpy()
    enc_lens = ((lens + 2) // 3).numpy()

    lat_events = []
    _native.EVENT_HOOK = None

    # the hooks of the recipe that touch gradients / parameters every step
    # (egs/wsj/yamls/ctc.yaml:90-103): global-norm clipping incl. skip-step (it
    # sees the all-reduced gradient) and the Polyak average of the state_dict.
    # The recipe's thresholds (clip 1e4, skip 1e5) are for its global batch of 16
    # utterances and a loss SUMMED over utterances: they are scaled with the
    # global batch / 16 here, otherwise every step of a large batch is skipped.
    hooks = []
    clip_scale = B * world / 16.0
    if not a.no_hooks:
        hooks = [GradientClipping(clip_norm=10000.0 * clip_scale,
                                  skip_step_norm=100000.0 * clip_scale),
                 PolyakDecay(decay_rates=[0.9998])]
        for h in hooks:
            h.pre_run(model, opt)
    # the step boundary on the device (att_speech/fused_step.py): same clip / skip rule and Adam
    # arithmetic as the hook + torch.optim.Adam, no read-back between backward and the update
    fused = None
    if hooks and not a.host_step:
        from att_speech.fused_step import FusedClipAdam
        fused = FusedClipAdam.from_optimizer(opt, bucket, hooks[0])

    class _Recorder(object):                 # times the lattice call of the recorded steps
        def __init__(self):
            self.on = False

        def __call__(self, *args, **kw):
            _native.EVENT_HOOK = lat_events if self.on else None
            try:
                return model(*args, **kw)
            finally:
                _native.EVENT_HOOK = None
    fwd = _Recorder()
    skipped = []

    def step(record=False):
        fwd.on = record
        with contextlib.redirect_stdout(sys.stderr):     # hooks print like the reference; stdout is the JSON line
            out, skip = train_step(model, opt, ((feats_d, lens, None, texts, llens), {}),
                                   hooks=hooks, bucket=bucket, forward=fwd, fused=fused)
        if record:
            skipped.append(bool(skip))
        return out['loss']

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t_start = time.time()

    def progress(msg):
        if rank == 0:
            print('[bench %6.1fs] %s' % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    loss0 = None
    for i in range(a.warmup):
        l = step()
        torch.cuda.synchronize()
        if loss0 is None:
            loss0 = float(l.detach())
        progress('warmup step %d done' % i)
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        loss = step(record=True)
    fence()
    dt = time.time() - t0
    if fused is not None:        # the device's decisions of the timed steps, read after the fence
        skipped = [s_ or rec[2] for s_, rec in zip(skipped, fused.drain()[-a.steps:])]
        assert fused.steps_taken == a.warmup + a.steps - sum(fused_rec[2] for fused_rec in fused.history)
    # every timed step must have taken its optimizer step (work skipped inside the
    # timed region would invalidate the number)
    assert not any(skipped), 'optimizer step skipped in %d of %d timed steps' % (sum(skipped), len(skipped))
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    frames = torch.tensor([float(lens.sum())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    total_frames = float(frames.item())
    lstm_err = getattr(_native, 'lstm_check_errors', None)
    if lstm_err is not None:
        lstm_err()                               # raises if a persistent-LSTM hand-off timed out

    # BASELINE configs 3 and 5 beside the headline (every rank takes part when N > 1)
    extra = []
    if not a.no_extra and a.workload in (None, 'ctc') and order == 1:
        final_state = model.state_dict() if rank == 0 else None
        del feats_d, bucket, opt
        torch.cuda.empty_cache()
        # (the batch of the headline: the persistent LSTM grid is sized for 768 utterances per GPU,
        # and the wide-alphabet decoders scale with the batch — ctc_bi 23.3 M frames/s at 512
        # utterances, 25.5 M at 768; ctcg_bi_cde 14.8 M at 256, 19.4 M at 512, 21.1 M at 768)
        for wl, wb in ([('ctc_bi', 768), ('ctcg_bi_cde', 768)] if world == 1 else [('ctc_bi', 768)]):
            progress('extra workload %s' % wl)
            extra.append(time_extra_workload(wl, wb, T, 5, 2, dev, rank, world, not a.no_hooks,
                                             not a.host_step))
    else:
        final_state = model.state_dict() if rank == 0 else None

    if rank == 0:
        lat_ms = [s.elapsed_time(e) for (s, e) in lat_events]
        lat_ms = float(np.mean(lat_ms)) if lat_ms else float('nan')
        alg = lattice_algorithmic_bytes(enc_lens, C, llens.numpy(), n_arcs)
        achieved = alg / (lat_ms * 1e-3) / 1e9
        flops = step_flops(B, T, C)
        tflops = flops * world * a.steps / dt / 1e12
        hook_txt = '' if a.no_hooks else ('GradientClipping(clip %g, skip %g = recipe x global_batch/16)'
                                          '+PolyakDecay+' % (1e4 * clip_scale, 1e5 * clip_scale))
        res = {
            'metric': 'CTC train-step frames/sec',
            'value': total_frames * a.steps / dt,
            'unit': 'frames/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'WSJ %s-char CTC (egs/wsj/yamls/%s.yaml shapes), DeepSpeech2 '
                                   'conv+4xBiLSTM-320 encoder + FSTDecoder, fwd+bwd+%sAdam, '
                                   'synthetic %d-dim x %d-channel x %d-frame fbank'
                                   % ('mono' if order == 1 else 'bi',
                                      a.workload or ('ctc' if order == 1 else 'ctc_bi'),
                                      hook_txt, FEATS[0], FEATS[1], T),
                       'batch_per_gpu': B, 'global_batch': B * world, 'frames': T,
                       'classes': C, 'parallelism': 'dp%d' % world,
                       'batch_norm': 'statistics over all replicas' if a.sync_bn else 'per-replica statistics',
                       'optimizer_steps': len(skipped) - sum(skipped),
                       'step_boundary': ('device: att_speech.fused_step (clip / skip decision + Adam, no read-back)'
                                         if fused is not None else 'host: hook + torch.optim.Adam, one read-back'),
                       'first_loss': loss0, 'final_loss': float(loss.detach())},
            'roofline': {'bound': 'hbm', 'kernel': 'lattice_fwbw (alpha/beta scan): %s' % (
                             mono_lattice_kernel(B) if order == 1 else 'lattice_fwbw_sl_kernel<3, 8, 0>'),
                         'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': pmc_traffic(order, B, T),
                         'algorithmic_bytes_per_launch': alg,
                         'avg_launch_ms': lat_ms},
            'roofline_mfma': {'bound': 'mfma', 'kernel': 'whole step (conv + BiLSTM + projection, fwd+bwd)',
                              'achieved': tflops, 'peak': MFMA_PEAK_TFLOPS * world, 'unit': 'TFLOP/s',
                              'frac': tflops / (MFMA_PEAK_TFLOPS * world),
                              'flops_per_step_per_gpu': flops},
        }
        if world == 1 and not a.no_extra:
            progress('bi-char numerator roofline launch')
            res['roofline_bichar'] = bichar_numerator_roofline(
                dev, traffic=pmc_traffic(2, B, T, 'lattice_fwbw_sl_kernel<3, 8, 0>', grid=512 * 512))
            torch.cuda.empty_cache()
            progress('stage-2 decode rate')
            res['decode'] = tcn_decode_rate(dev)
        res['extra_workloads'] = extra
        if world == 1 and not a.no_cpu_baseline and a.workload != 'ctcg_bi_cde':
            progress('loss delta + timing the CPU baseline (about 25 s)')
            res['cpu_baseline'], res['loss_delta'] = cpu_baseline(T, order, dev, final_state)
        else:
            res['cpu_baseline'] = None
            res['loss_delta'] = None
        print(json.dumps(res), file=_STDOUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries exactly ONE line (the JSON); library / hook chatter goes to stderr
    _STDOUT = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        main()
