"""Summarise two rocprofv3 PMC passes (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each
with `--kernel-trace --output-format csv`) into profiles/*.json: mean counter value per
dispatch and kernel, in KB as rocprofv3 reports them.  bench.py reads the lattice
kernel's entry for `roofline.traffic`.

  python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> --batch 768
"""
import argparse
import collections
import csv
import glob
import json
import os


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir'); ap.add_argument('write_dir'); ap.add_argument('out')
    ap.add_argument('--batch', type=int, required=True)
    ap.add_argument('--cmd', default='bench.py --steps 2 --warmup 1 --no-cpu-baseline')
    a = ap.parse_args()
    fe, wr = collect(a.fetch_dir, 'FETCH_SIZE'), collect(a.write_dir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        name = k[:100]
        kernels[name] = {'dispatches': len(fe.get(k, wr.get(k))),
                         'fetch_KB': round(sum(fe[k]) / len(fe[k]), 1) if k in fe else None,
                         'write_KB': round(sum(wr[k]) / len(wr[k]), 1) if k in wr else None}
    json.dump({'note': 'separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `%s` (B=%d); raw '
                       'counter means per dispatch in KB. On gfx950 FETCH_SIZE under-reports '
                       'coalesced 4 B/lane streams by 2x (calibration in DESIGN.md §5); WRITE_SIZE '
                       'is exact.' % (a.cmd, a.batch),
               'batch': a.batch, 'kernels': kernels}, open(a.out, 'w'), indent=1)
    print('wrote', a.out, len(kernels), 'kernels')


if __name__ == '__main__':
    main()
