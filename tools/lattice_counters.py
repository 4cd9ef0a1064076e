"""Launches the two mono-char CTC lattice kernels (log-domain state-labelled kernel and the
linear-domain band kernel) on the bench's lattice shape, for a counter pass:
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/lattice_counters.py
tools/pmc_counters.py summarises the passes (profiles/r03_pmc_lattice_issue.json)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-asr_amd'), os.path.join(ROOT, 'tools')]
from att_speech import _native     # noqa: E402
import bench_lattice               # noqa: E402


def main():
    dev = torch.device('cuda:0')
    T = 334
    for B in (256, 512, 768):
        lens, mats, C, n_states, n_arcs = bench_lattice.make(1, B, T, 'num')
        g = _native.Graph(mats, dev)
        lp = _native.log_softmax_fwd(torch.randn(T, B, C, device=dev), C)
        tl = torch.from_numpy(lens).to(dev)
        for band in ('0', '2'):
            os.environ['ASR_LATTICE_BAND'] = band
            for _ in range(4):
                _native.lattice_fwbw(lp, tl, g)
            torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
