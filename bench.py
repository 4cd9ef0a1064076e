#!/usr/bin/env python
"""bench.py — CTC train-step frames/s on MI355X (BASELINE.json metric).

One "step" = forward + backward + Adam of SpeechModel(DeepSpeech2 encoder +
FSTDecoder, mono-char CTC: egs/wsj/yamls/ctc.yaml shapes) on one synthetic batch
of B utterances x 1000 fbank frames x 40 dims per GPU; frames/s is the reference's
own throughput definition, sum(feature_lens) / step_time (trainer.py:292-298).
N > 1: one process per GPU (torch.distributed.run), utterance-sharded data
parallelism, one RCCL all-reduce of the flat gradient bucket per step.

Prints ONE JSON line on rank 0 (see the contract in the task statement), with
  roofline     — the lattice forward-backward scan kernel, HBM-bound, timed live
                 with events on the stream it is launched on;
  cpu_baseline — the same step on the host cores (torch-CPU encoder + the CPU
                 oracle for the lattice), bounded sample, N = 1 only.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# pinned MIOpen solver choice for the conv layers that still go through torch
# (see pytorch-asr_amd/miopen_db/README.md); must be set before MIOpen loads
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'pytorch-asr_amd', 'miopen_db'))


def pin_gemm_selection(local_rank):
    """Library GEMM solutions measured fastest for this step's bf16 GEMM shapes (PyTorch
    TunableOp results committed under pytorch-asr_amd/tunableop/, tuning OFF): the same
    kind of pin as the MIOpen find-db above.  TunableOp reads `<name><device>.csv`, so
    every rank gets its own copy (in its own temporary directory).  Any PYTORCH_TUNABLEOP_* setting of the caller wins."""
    src = os.path.join(ROOT, 'pytorch-asr_amd', 'tunableop', 'gfx950_b576.csv')
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


def synthetic_batch(B, T, rank, order=1):
    """features N(0,1) [B,T,40,1]; saturating batch = all-T
    lengths (B > 16), YAML batch = T - 8b; labels uniform in [2,48],
    L_b = 100 - 2 (b mod 16)."""
    g = torch.Generator().manual_seed(1234 + rank)
    feats = torch.randn(B, T, 40, 1, generator=g)
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


def cpu_baseline(T, order, seconds_budget=15.0):
    """The same training step on the host: torch-CPU encoder/projection +
    oracle/lattice_oracle.c for the lattice (kind 'port'), small batch."""
    from att_speech.models import SpeechModel
    from att_speech import fst_utils
    from oracle import oracle

    class OracleLattice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, lens, mats):
            r = oracle.path_logsumexp(lp.detach().numpy(), lens.numpy(),
                                      [m.numpy() for m in mats])
            ctx.grads = torch.from_numpy(r['grad'])
            return torch.from_numpy(r['logZ'])

        @staticmethod
        def backward(ctx, g):
            return g[None, :, None] * ctx.grads, None, None

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

    def step():
        opt.zero_grad()
        enc, elens = model.encoder(feats, lens, None)
        logits = model.decoder.fc(enc)
        lp = torch.log_softmax(logits, -1)
        mx = lp.max(-1, keepdim=True)[0].detach()
        mask = (torch.arange(lp.size(0))[:, None] < elens[None, :]).float()
        mats = gg.get_training_matrices_batch(texts, llens)
        num = -OracleLattice.apply(lp - mx, elens, mats)
        loss = (num - (mx.squeeze(-1) * mask).sum(0)).sum()
        loss.backward()
        opt.step()
        return float(loss)

    step()                                   # warm-up
    t0, n = time.time(), 0
    while n < 2 or (time.time() - t0 < seconds_budget and n < 50):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=float(lens.sum()) / dt, unit='frames/s', cores=cores, kind='port',
                sample='%d steps of the same train step at B=%d x %d frames on the host '
                       '(torch-CPU encoder + oracle/lattice_oracle.c lattice, 1 thread)'
                       % (n, B, T))


def pmc_traffic(order, B, T):
    """HBM bytes per lattice launch from the PMC passes committed under profiles/
    (r01_pmc_step_fetch_write.json / r01_pmc_lattice_fetch_write.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of bench.py (tools/pmc_summary.py) / tools/bench_lattice.py, T'=334; FETCH_SIZE reads half of a 4 B/lane coalesced
    stream on gfx950 - calibrated on log_softmax_fwd - so traffic = 2*FETCH + WRITE).
    bench.py cannot collect counters itself; the figure applies to the measured shape only."""
    if T != 1000:
        return None
    k = 'lattice_fwbw_sl_kernel<3, 8, 1>' if order == 1 else 'lattice_fwbw_sl_kernel<3, 8, 0>'
    try:    # passes over bench.py itself (the lattice launch inside the training step)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_step_fetch_write.json')))
        if pmc.get('batch', 512) != B:
            return None
        e = next(v for n, v in pmc['kernels'].items() if k in n)
        return (2 * e['fetch_KB'] + e['write_KB']) * 1024.0
    except (OSError, KeyError, ValueError, StopIteration, TypeError):
        pass
    if B != 512:
        return None
    try:    # passes over tools/bench_lattice.py (same shapes, kernel alone)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_lattice_fetch_write.json')))
        return (2 * pmc['FETCH_SIZE'][k]['mean_KB'] + pmc['WRITE_SIZE'][k]['mean_KB']) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=576,
                    help='utterances per GPU (576 = 24 batch tiles of 24 rows: the persistent LSTM '
                         'grid of 48 teams x 5 workgroups exactly)')
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--order', type=int, default=1, help='1 mono-char CTC, 2 bi-char CTC')
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help='BASELINE config; default ctc (mono-char, the headline metric)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hooks', action='store_true',
                    help='leave out GradientClipping / PolyakDecay (recipe hooks)')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    pin_gemm_selection(local_rank)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus

    from att_speech import _native, fst_utils
    from att_speech.dp import FlatGradBucket, broadcast_parameters, train_step
    from att_speech.modules.hooks import GradientClipping, PolyakDecay
    from att_speech.models import SpeechModel

    if a.workload:
        a.order = WORKLOADS[a.workload][0]
    B, T, order = a.batch, a.frames, a.order
    C = S ** order
    feats, lens, texts, llens = synthetic_batch(B, T, rank, order)
    enc_cfg, dec_cfg = model_config(order, a.workload)
    torch.manual_seed(1234)
    sb = {'features': feats[:2].clone(), 'features_lengths': lens[:2].clone(), 'spkids': None}
    model = SpeechModel(enc_cfg, dec_cfg, sb, C, [str(i) for i in range(S)]).to(dev)
    broadcast_parameters(model)
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
    # sees the all-reduced gradient) and the Polyak average of the state_dict
    hooks = []
    if not a.no_hooks:
        hooks = [GradientClipping(clip_norm=10000.0, skip_step_norm=100000.0),
                 PolyakDecay(decay_rates=[0.9998])]
        for h in hooks:
            h.pre_run(model, opt)

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

    def step(record=False):
        fwd.on = record
        with contextlib.redirect_stdout(sys.stderr):     # hooks print like the reference; stdout is the JSON line
            out, _ = train_step(model, opt, ((feats_d, lens, None, texts, llens), {}),
                                hooks=hooks, bucket=bucket, forward=fwd)
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

    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        progress('warmup step %d done' % i)
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        loss = step(record=True)
    fence()
    dt = time.time() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    frames = torch.tensor([float(lens.sum())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    total_frames = float(frames.item())

    if rank == 0:
        lat_ms = [s.elapsed_time(e) for (s, e) in lat_events]
        lat_ms = float(np.mean(lat_ms)) if lat_ms else float('nan')
        alg = lattice_algorithmic_bytes(enc_lens, C, llens.numpy(), n_arcs)
        achieved = alg / (lat_ms * 1e-3) / 1e9
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
                                   'synthetic 40-dim x %d-frame fbank'
                                   % ('mono' if order == 1 else 'bi',
                                      a.workload or ('ctc' if order == 1 else 'ctc_bi'),
                                      '' if a.no_hooks else 'GradientClipping+PolyakDecay+', T),
                       'batch_per_gpu': B, 'global_batch': B * world, 'frames': T,
                       'classes': C, 'parallelism': 'dp%d' % world,
                       'final_loss': float(loss.detach())},
            'roofline': {'bound': 'hbm', 'kernel': 'lattice_fwbw (alpha/beta scan)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': pmc_traffic(order, B, T),
                         'algorithmic_bytes_per_launch': alg,
                         'avg_launch_ms': lat_ms},
        }
        if world == 1 and not a.no_cpu_baseline and a.workload != 'ctcg_bi_cde':
            progress('timing the CPU baseline (about 20 s)')
            res['cpu_baseline'] = cpu_baseline(T, order)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res), file=_STDOUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries exactly ONE line (the JSON); library / hook chatter goes to stderr
    _STDOUT = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        main()
