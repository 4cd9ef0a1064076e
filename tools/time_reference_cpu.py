#!/usr/bin/env python
"""Times the REFERENCE's own lattice code on this container's CPU cores.

Runs ONLY in the build container, where /root/reference exists: it imports the reference's
`att_speech.fst_utils` exactly as tests/golden/make_golden.py does (py3.10, empty stub modules
for the absent pywrapfst / torchtext / kaldi_io / tensorboardX) and times
`path_reduction(..., red_kind='logsumexp')` -> `PathLogSumExp` (fst_utils.py:322-488), forward
plus the backward through it, fp32, `torch.set_num_threads(<cores>)`, on
  * the mono-character numerator lattice of the bench workload (T' = 334, C = 49, L <= 100),
  * the bi-character numerator lattice (C = 2401, contextual blanks),
graphs from this repo's graph builder (OpenFst is absent; the matrices are checked against
oracle/fst_oracle.py in tests/test_graphs.py).  Prints a markdown table; `--write` replaces the
table between the markers in BASELINE.md §2.  A reported baseline, not a target.
"""
import argparse
import os
import re
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'


def import_reference():
    if not os.path.isdir(REF):
        sys.exit('tools/time_reference_cpu.py: %s is not here (build container only)' % REF)
    for name in ['pywrapfst', 'torchtext', 'torchtext.vocab', 'kaldi_io', 'tensorboardX']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['torchtext'].vocab = sys.modules['torchtext.vocab']
    sys.modules['torchtext.vocab'].Vocab = object
    sys.modules['tensorboardX'].SummaryWriter = object
    sys.path.insert(0, REF)
    from att_speech import fst_utils as ref_fst
    return ref_fst


def our_graphs(order, B, S=49):
    """numerator lattices of the bench workload from THIS repo's builder (a separate
    interpreter state would be cleaner; the two `att_speech` packages cannot be imported
    together, so the matrices are built in a child process and handed over as arrays)"""
    import subprocess
    import numpy as np
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
from att_speech import fst_utils
rng = np.random.default_rng(1234)
B, S, order = %d, %d, %d
llens = np.array([100 - 2 * (b %% 16) for b in range(B)])
labs = rng.integers(2, S, size=(B, 100))
if order == 2:
    prev = np.concatenate([np.zeros((B, 1), labs.dtype), labs[:, :-1]], 1)
    labs = prev * S + labs
mats = fst_utils.CTCGraphGen(context_order=order, num_symbols=S).get_training_matrices_batch(labs, llens)
np.savez(sys.argv[1], *[m.numpy() for m in mats])
''' % (os.path.join(ROOT, 'pytorch-asr_amd'), ROOT, B, S, order)
    path = '/tmp/ref_time_graph_%d_%d.npz' % (order, B)
    subprocess.check_call([sys.executable, '-c', code, path])
    z = np.load(path)
    return [z['arr_%d' % i] for i in range(len(z.files))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--write', action='store_true')
    a = ap.parse_args()
    mats_by_order = {o: our_graphs(o, a.batch) for o in (1, 2)}
    ref_fst = import_reference()
    import numpy as np
    import torch
    torch.set_num_threads(a.threads)
    T, B, S = 334, a.batch, 49
    rows = []
    for order, name in ((1, 'mono-char numerator lattice'), (2, 'bi-char numerator lattice (contextual blanks)')):
        C = S ** order
        mats = [torch.from_numpy(np.ascontiguousarray(m)) for m in mats_by_order[order]]
        N, K = mats[0].shape[1], mats[0].shape[2]
        g = torch.Generator().manual_seed(order)
        lens = torch.tensor([T - 7 * b for b in range(B)], dtype=torch.int32)
        best = None
        for _ in range(a.reps + 1):
            lp = torch.log_softmax(torch.randn(T, B, C, generator=g), -1).requires_grad_()
            t0 = time.perf_counter()
            logz = ref_fst.path_reduction(lp, lens, mats, red_kind='logsumexp', neg_inf=-1e20)
            logz.sum().backward()
            dt = time.perf_counter() - t0
            best = dt if best is None or _ == 0 else min(best, dt)      # first run = warm-up
        frames = int(lens.sum()) * 3
        rows.append('| %s | `T\'=%d, B=%d, C=%d, N=%d, K=%d`, lengths %d…%d | %.3f s | %.3g |' % (
            name, T, B, C, N, K, int(lens.max()), int(lens.min()), best, frames / best))
    table = '\n'.join([
        '| case (reference `PathLogSumExp` forward + backward, fp32, %d threads) | shapes | time / batch | input frames/s (×3 subsampling) |' % a.threads,
        '|---|---|---|---|'] + rows)
    print(table)
    if a.write:
        p = os.path.join(ROOT, 'BASELINE.md')
        s = open(p).read()
        m0, m1 = '<!-- time_reference_cpu:begin -->', '<!-- time_reference_cpu:end -->'
        block = '%s\n%s\n%s' % (m0, table, m1)
        if m0 in s:
            s = re.sub(re.escape(m0) + '.*?' + re.escape(m1), lambda _: block, s, flags=re.S)
        else:
            sys.exit('BASELINE.md has no %s marker' % m0)
        open(p, 'w').write(s)
        print('BASELINE.md §2 updated')


if __name__ == '__main__':
    main()
