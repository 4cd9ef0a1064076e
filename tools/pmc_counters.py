"""Mean counter values per (kernel, launch size) over any number of rocprofv3 --pmc passes
(`--kernel-trace --output-format csv` each), with the mean kernel duration beside them.
  python tools/pmc_counters.py <out.json> <match> <pass_dir> [<pass_dir> ...]
Only kernels whose name contains <match> are kept."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, match, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if match in r['Kernel_Name']:
                    key = (r['Kernel_Name'][:80], int(r.get('Grid_Size') or 0))
                    vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if match in r['Kernel_Name']:
                    key = (r['Kernel_Name'][:80], int(r.get('Grid_Size') or 0))
                    try:
                        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
                    except (KeyError, ValueError):
                        pass
    res = []
    for key in sorted(vals):
        e = {'kernel': key[0], 'grid_work_items': key[1],
             'avg_us_under_counters': round(sum(dur[key]) / max(1, len(dur[key])), 1)}
        for c, v in sorted(vals[key].items()):
            e[c] = round(sum(v) / len(v), 1)
        res.append(e)
    json.dump({'note': 'means per dispatch; SQ_* counters are summed over the chip',
               'kernels': res}, open(out, 'w'), indent=1)
    for e in res:
        print(json.dumps(e))


if __name__ == '__main__':
    main()
