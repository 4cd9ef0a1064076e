"""MFMA-busy fraction per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
--kernel-trace --output-format csv` pass of bench.py:
    busy cycles (summed over the chip's 1024 SIMDs) / (kernel duration x 2.4 GHz x 1024).
  python tools/mfma_busy.py <pmc_dir> <out.json> [--min-us 50]
"""
import argparse, collections, csv, glob, json, os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('pmc_dir'); ap.add_argument('out')
    ap.add_argument('--min-us', type=float, default=50.0)
    ap.add_argument('--cmd', default='bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra')
    a = ap.parse_args()
    busy = collections.defaultdict(list)
    for f in glob.glob(os.path.join(a.pmc_dir, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                busy[r['Kernel_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(a.pmc_dir, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    rows = []
    for k, v in busy.items():
        if k not in dur:
            continue
        us = sum(dur[k]) / len(dur[k])
        b = sum(v) / len(v)
        if us < a.min_us or b <= 0:
            continue
        rows.append({'kernel': k[:110], 'calls': len(v), 'avg_us_under_pmc': round(us, 1),
                     'mfma_busy_simd_cycles': b, 'mfma_busy_frac': round(b / (us * 1e-6 * 2.4e9 * 1024), 3)})
    rows.sort(key=lambda r: -r['avg_us_under_pmc'] * r['calls'])
    json.dump({'note': 'SQ_VALU_MFMA_BUSY_CYCLES (one --pmc pass of `%s`, summed over the 1024 SIMDs) '
                       '/ (the kernel\'s duration in the same pass x 2.4 GHz x 1024 SIMDs); durations under '
                       'the counter pass run a few per cent long' % a.cmd,
               'kernels': rows}, open(a.out, 'w'), indent=1)
    for r in rows[:16]:
        print('%6.1f us x%3d  busy %.3f  %s' % (r['avg_us_under_pmc'], r['calls'], r['mfma_busy_frac'], r['kernel'][:70]))


if __name__ == '__main__':
    main()
