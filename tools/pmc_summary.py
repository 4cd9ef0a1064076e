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
    """kernel name -> [(grid size in work-items, counter value)] over the dispatches"""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append((int(r.get('Grid_Size') or 0), float(r['Counter_Value'])))
    return acc


def mean(pairs, grid=None):
    v = [x for g, x in pairs if grid is None or g == grid]
    return round(sum(v) / len(v), 1) if v else None


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
                         'fetch_KB': mean(fe[k]) if k in fe else None,
                         'write_KB': mean(wr[k]) if k in wr else None}
        # one kernel launched on several problem sizes (the extra workloads of bench.py): the
        # means above mix them; by_grid keeps them apart (key = work-items of the launch)
        grids = sorted({g for g, _ in fe.get(k, [])} | {g for g, _ in wr.get(k, [])})
        if len(grids) > 1:
            kernels[name]['by_grid'] = {
                str(g): {'dispatches': sum(1 for gg, _ in fe.get(k, wr.get(k)) if gg == g),
                         'fetch_KB': mean(fe.get(k, []), g), 'write_KB': mean(wr.get(k, []), g)}
                for g in grids}
    json.dump({'note': 'separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `%s` (B=%d); raw '
                       'counter means per dispatch in KB. On gfx950 FETCH_SIZE under-reports '
                       'coalesced 4 B/lane streams by 2x (calibration in DESIGN.md §5); WRITE_SIZE '
                       'is exact.' % (a.cmd, a.batch),
               'batch': a.batch, 'kernels': kernels}, open(a.out, 'w'), indent=1)
    print('wrote', a.out, len(kernels), 'kernels')


if __name__ == '__main__':
    main()
