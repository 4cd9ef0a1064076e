"""Host-side pieces of bench.py and tools/pmc_summary.py (no GPU): the algorithmic-bytes
formula of SURVEY.md §8d, the synthetic batch of BASELINE.md §3, the PMC traffic lookup and
the GEMM-selection pin."""
import csv
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_and_synthetic_batch():
    b = load(os.path.join(ROOT, 'bench.py'), 'bench_mod')
    # one utterance, T'=10, C=5, L=3 (N=7), E=17 arcs: 4*10*(15+14) + 24*17 + 28
    assert b.lattice_algorithmic_bytes([10], 5, [3], [17]) == 4 * 10 * 29 + 24 * 17 + 28
    feats, lens, texts, llens = b.synthetic_batch(32, 1000, 0, 1)
    assert feats.shape == (32, 1000, 40, 1) and lens.tolist() == [1000] * 32
    assert llens.tolist() == [100 - 2 * (i % 16) for i in range(32)]
    assert int(texts.max()) <= 48 and int(texts[0, :100].min()) >= 2
    assert int(texts[1, 98:].abs().sum()) == 0                      # padding past L_b
    f2, _, t2, _ = b.synthetic_batch(32, 1000, 0, 2)                # bigram ids prev*49+cur
    assert (feats == f2).all() and int(t2[0, 0]) == int(texts[0, 0])
    assert int(t2[0, 5]) == int(texts[0, 4]) * 49 + int(texts[0, 5])
    _, yl, _, _ = b.synthetic_batch(16, 1000, 0, 1)                  # the YAML batch
    assert yl.tolist() == [1000 - 8 * i for i in range(16)]


def test_pmc_traffic_reads_the_committed_summary():
    b = load(os.path.join(ROOT, 'bench.py'), 'bench_mod2')
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r03_pmc_step_fetch_write.json')))
    e = next(v for k, v in pmc['kernels'].items() if b.mono_lattice_kernel(pmc['batch']) in k)
    want = (2 * e['fetch_KB'] + e['write_KB']) * 1024.0        # FETCH_SIZE x2: DESIGN.md §5
    assert b.pmc_traffic(1, pmc['batch'], 1000) == want
    assert b.pmc_traffic(1, pmc['batch'] + 1, 1000) is None     # measured shape only
    assert b.pmc_traffic(1, pmc['batch'], 999) is None


def test_gemm_pin_respects_the_caller():
    b = load(os.path.join(ROOT, 'bench.py'), 'bench_mod3')
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith('PYTORCH_TUNABLEOP_')}
    try:
        b.pin_gemm_selection(3)
        assert os.environ['PYTORCH_TUNABLEOP_ENABLED'] == '1'
        assert os.environ['PYTORCH_TUNABLEOP_TUNING'] == '0'
        name = os.environ['PYTORCH_TUNABLEOP_FILENAME']
        copy = name[:-len('.csv')] + '3.csv'                      # TunableOp appends the device
        assert open(copy).read() == open(os.path.join(
            ROOT, 'pytorch-asr_amd', 'tunableop', 'gfx950.csv')).read()
        for k in [k for k in os.environ if k.startswith('PYTORCH_TUNABLEOP_')]:
            del os.environ[k]
        os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '0'             # any setting of the caller wins
        b.pin_gemm_selection(0)
        assert 'PYTORCH_TUNABLEOP_FILENAME' not in os.environ
    finally:
        for k in [k for k in os.environ if k.startswith('PYTORCH_TUNABLEOP_')]:
            del os.environ[k]
        os.environ.update(saved)


def test_pmc_summary_tool(tmp_path, monkeypatch):
    t = load(os.path.join(ROOT, 'tools', 'pmc_summary.py'), 'pmc_summary')
    for d, counter, vals in (('f', 'FETCH_SIZE', [10.0, 14.0]), ('w', 'WRITE_SIZE', [5.0, 7.0])):
        os.makedirs(tmp_path / d / 'host')
        with open(tmp_path / d / 'host' / '1_counter_collection.csv', 'w') as f:
            w = csv.writer(f)
            w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
            for i, v in enumerate(vals):
                w.writerow([i, 'kern_a(int)', counter, v])
            w.writerow([9, 'kern_a(int)', 'OTHER', 1.0])
    out = tmp_path / 'o.json'
    monkeypatch.setattr(sys, 'argv', ['pmc_summary.py', str(tmp_path / 'f'), str(tmp_path / 'w'),
                                      str(out), '--batch', '576'])
    t.main()
    j = json.load(open(out))
    assert j['batch'] == 576
    assert j['kernels']['kern_a(int)'] == {'dispatches': 2, 'fetch_KB': 12.0, 'write_KB': 6.0}


def test_step_flops_matches_survey_figures():
    """LSTM 13.27 MFLOP + conv2 1.10 MFLOP + projection 0.03 MFLOP per encoded
    frame forward (synthetic shape), conv1 2*32*17*49 per conv1 output frame; training = 3x."""
    b = load(os.path.join(ROOT, 'bench.py'), 'bench_mod4')
    B, T, C = 2, 1000, 49
    per_enc = 13.27e6 + 2 * 32 * 11 * 32 * 49 + 2 * 320 * C
    want = 3 * B * (334 * per_enc + 1006 * 2 * 32 * 17 * 49)
    assert abs(b.step_flops(B, T, C) - want) < 2e-3 * want


def test_gpus_n_without_rank_environment_launches_the_ranks():
    """`python bench.py --gpus 2` the way the driver calls it: the process starts two ranks
    itself (torch.distributed.run child), relays ONE JSON line and the child's exit code.
    Rehearsed without a GPU (gloo ranks, no model)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3',
                        '--warmup', '1', '--batch', '4', '--dry-run-launcher'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300,
                       universal_newlines=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['dry_run'] is True
    # a world size that disagrees with --gpus is an error exit, not an assert deep inside
    env2 = dict(env, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-launcher'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env2, timeout=120,
                       universal_newlines=True)
    assert r.returncode != 0 and r.stdout.strip() == ''
